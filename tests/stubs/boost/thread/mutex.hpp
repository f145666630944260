#pragma once
namespace boost { struct mutex {}; }
