#pragma once
namespace std_srvs { struct Empty { struct Request {}; struct Response {}; }; }
