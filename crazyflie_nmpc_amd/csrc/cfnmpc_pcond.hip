// cfnmpc_pcond.hip -- partial condensing path of the batched RTI step (cfnmpc_opts.cond_N2 < N).
//
// Role in the reference: PARTIAL_CONDENSING_HPIPM (generate_c_code.py:140, README.md:77) -- HPIPM's
// d_part_cond / d_ocp_qp_ipm / expand, none of which is under /root/reference (empty acados
// submodule); this file implements the published algorithm from the mathematics (the tests hold a
// numpy restatement -- partial_condense / riccati_condensed / expand_condensed -- as the checker):
//
//   k_pcond   : the N stages are regrouped into N2 blocks of m consecutive stages.  With
//               z = (dU, dx, 1), dU = the 4 m inputs of the block, dx = the state at its start:
//                   dx_{k0+i} = G_i z,   G_0 = [0 I 0],   G_{i+1} = A_{k0+i} G_i + [B at du_i | b at 1]
//               the block's cost is 1/2 z'H z with H = sum_i G_i'Q~_i G_i + R, r terms and its
//               dynamics dx_{k0+m} = D z, D = G_m.  One (instance, block) per lane group; blocks are
//               independent (grid = instances x N2).
//   k_cfactor : Riccati recursion over the N2 condensed stages, backward: H~ = H + E'P~E with
//               E = [D; e_1] and the augmented cost-to-go P~ (13 x 13 matrix + affine row), ONE dense
//               Cholesky factorisation of the 4m x 4m input block per stage (right-looking in panels of
//               four columns, in LDS), gains K (4m x 13) and feed-forward d by panel-wise
//               back-substitution, P~ <- Schur complement.
//               The gains are stored per ORIGINAL stage in the layout of the uncondensed path.
//   k_forward<COND> (cfnmpc_kernels.hip): forward sweep + `expand`: the inputs of a block are the
//               condensed feedback law at the state of the block's START, the interior states follow
//               from the original (matrix-free RK4) stage dynamics.
//   k_cipm    : instances whose unconstrained minimiser leaves the input box: Mehrotra
//               predictor-corrector interior point in delta form on the SAME condensed blocks (the
//               barrier only changes the diagonal of the input block and the gradient), two
//               factorisations per iteration; expand through the stored (A, B, b).
//
// Mapping: one instance per group of LPI lanes -- the code is generic in LPI, the launchers use 64 (one
// instance per wavefront: measured 3-5x faster than 16 lanes per instance, since the LDS footprint per
// INSTANCE fixes how many instances a CU holds either way) --, all dense blocks of an instance in LDS,
// the block length m is a run-time value up to the template's MMAX (leading dimensions are the template's).
// Global loads are batched and issued ahead everywhere: a load inside a run-time loop costs one HBM round
// trip per iteration, which dominated the first version of these kernels.  This path is an
// OPTION (parity with the reference's solver plan + the N2 sweep of DESIGN.md section 5.8), not the
// default: condensing raises both the bytes per stage and the flops for this problem's sizes.
#include <hip/hip_runtime.h>

#include "cfnmpc_model.hpp"
#include "cfnmpc_ws.hpp"

namespace cfn {
namespace {

typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) int gint;
__device__ __forceinline__ gdouble* gm(double* p) { return (gdouble*)(unsigned long long)p; }
__device__ __forceinline__ const gdouble* gm(const double* p) { return (const gdouble*)(unsigned long long)p; }
__device__ __forceinline__ gint* gm(int* p) { return (gint*)(unsigned long long)p; }

__device__ __forceinline__ int tri(int r, int c) { return (r * (r + 1)) / 2 + c; }   // c <= r
__device__ __forceinline__ int trs(int r, int c) { return r >= c ? tri(r, c) : tri(c, r); }
__device__ __forceinline__ double rsqrt_nr(double s) {
    double y = __builtin_amdgcn_rsq(s);
    const double hs = 0.5 * s;
    y = y * (1.5 - hs * y * y);
    y = y * (1.5 - hs * y * y);
    return y;
}
__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}

// One instance per group of LPI lanes.  Groups without an instance work on the spare workspace
// block (index NW: finite data, never read by anyone else).
template <int LPI>
struct Grp {
    int lane, g, inst, q;
    size_t wave;
    bool valid;
};
template <int LPI>
__device__ __forceinline__ Grp<LPI> grp_id(const Params& P, int first_inst) {
    Grp<LPI> t;
    t.lane = threadIdx.x % LPI;
    t.g = threadIdx.x / LPI;
    const int raw = first_inst + t.g;
    t.valid = raw < P.B;
    t.inst = t.valid ? raw : P.NW * 4 + (t.g & 3);
    t.wave = (size_t)(t.inst >> 2);
    t.q = t.inst & 3;
    return t;
}
template <int LPI>
__device__ __forceinline__ double grp_sum(double v) {
    for (int off = LPI / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
template <int LPI>
__device__ __forceinline__ double grp_min(double v) {
    for (int off = LPI / 2; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}
template <int LPI>
__device__ __forceinline__ double grp_max(double v) {
    for (int off = LPI / 2; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

// element (r, c) of A_k / B_k / b_k of an instance, internal state order, from the wave-blocked
// row-distributed stage blocks of the linearisation (cfnmpc_ws.hpp)
template <int LPI>
__device__ __forceinline__ double a_elem(const Params& P, const Grp<LPI>& t, int k, int r, int c) {
    if (c < 3) return r == c ? 1.0 : 0.0;
    const int s = c - 3;
    const int n = ar_n(s);
    if (r >= n) return 0.0;
    return gm(P.AR)[abidx(P, t.wave, k) * SZ_A + 4 * ar_pre(s) + t.q * n + r];
}
template <int LPI>
__device__ __forceinline__ double b_elem(const Params& P, const Grp<LPI>& t, int k, int r, int a) {
    return gm(P.BR)[abidx(P, t.wave, k) * SZ_B + (a * 4 + t.q) * 13 + r];
}
template <int LPI>
__device__ __forceinline__ double v13(const double* f, const Grp<LPI>& t, int stages, int k, int r) {
    return gm(f)[(t.wave * stages + k) * SZ_V13 + t.q * 13 + r];
}
template <int LPI>
__device__ __forceinline__ gdouble* cb_ptr(const Params& P, const Grp<LPI>& t, int j) {
    return gm(P.cb) + ((size_t)t.inst * P.cond_N2 + j) * cb_size(cond_mmax(P));
}

// ---------------------------------------------------------------------------------------------
// pcond: one (instance, block) per group
// ---------------------------------------------------------------------------------------------
// index of the row of entry e of a row-major lower triangle (e = r (r + 1) / 2 + c, c <= r)
__device__ __forceinline__ int tri_row(const int e) {
    int r = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    r += ((r + 1) * (r + 2)) / 2 <= e ? 1 : 0;
    r -= (r * (r + 1)) / 2 > e ? 1 : 0;
    return r;
}

template <int MMAX, int LPI>
__global__ __launch_bounds__(64) void k_pcond(Params P) {
    // LDS per group: H (packed lower, w x w), G and Gq = diag(Q) G (13 x W, leading dimension W = the
    // template's widest block: compile-time strides), the stage's dense A, B, b and gradient terms
    constexpr int W = cond_w(MMAX), IPW = 64 / LPI;
    constexpr int GSZ = cond_tri(W) + 2 * 13 * W + 169 + 52 + 13 + 13 + 4 + 13 + 4 + 3;
    constexpr int GST = (GSZ + 1) & ~1;
    __shared__ double lds[IPW * GST];
    const Grp<LPI> t = grp_id<LPI>(P, blockIdx.x * IPW);
    double* H = lds + t.g * GST;
    double* G = H + cond_tri(W);
    double* Gq = G + 13 * W;
    double* Am = Gq + 13 * W;
    double* Bm = Am + 169;
    double* bv = Bm + 52;
    double* qv = bv + 13;
    double* rv = qv + 13;
    double* wq = rv + 4;
    double* wr = wq + 13;
    const int j = blockIdx.y, N = P.N;
    const int m = cond_len(P, j), k0 = cond_start(P, j);
    const int mu = 4 * m, w = mu + 14, aff = w - 1;
    for (int e = t.lane; e < cond_tri(w); e += LPI) H[e] = 0.0;
    for (int e = t.lane; e < 13 * W; e += LPI) {
        const int r = e / W, c = e - r * W;
        G[e] = (c == mu + r) ? 1.0 : 0.0;
    }
    // structural entries of A (identity p-block, zeros) never change; the stored ones are refreshed per stage
    for (int e = t.lane; e < 169; e += LPI) Am[e] = (e / 13 == e % 13 && e % 13 < 3) ? 1.0 : 0.0;
    if (t.lane < 13) wq[t.lane] = P.W[ext_of(t.lane)];
    if (t.lane < 4) wr[t.lane] = P.W[13 + t.lane];
    // this lane's share of a stage's stored entries (decoded once): up to two of A's 97, one of B's 52
    int aoff[2], adst[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int e = t.lane + LPI * u;
        aoff[u] = -1; adst[u] = 0;
        if (e < 97) {
            int sl = 0;
            for (int s2 = 0; s2 < 10; s2++) if (e >= ar_pre(s2)) sl = s2;
            const int r = e - ar_pre(sl);
            aoff[u] = 4 * ar_pre(sl) + t.q * ar_n(sl) + r;
            adst[u] = r * 13 + sl + 3;
        }
    }
    constexpr int U = (cond_tri(W) + LPI - 1) / LPI;
    int pk[U];   // (row | column << 8) of this lane's entries of a row-major lower triangle
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int e = t.lane + LPI * u, r = tri_row(e);
        pk[u] = r | ((e - (r * (r + 1)) / 2) << 8);
    }
    const int boff = t.lane < 52 ? ((t.lane & 3) * 4 + t.q) * 13 + (t.lane >> 2) : -1;   // element (row lane >> 2, input lane & 3)
    double sa0, sa1, sb, sbv, sx, sy, su, syu;
    auto load_stage = [&](const int k) {
        const gdouble* ab = gm(P.AR) + abidx(P, t.wave, k) * SZ_A;
        sa0 = ab[max(aoff[0], 0)];
        sa1 = ab[max(aoff[1], 0)];
        sb = gm(P.BR)[abidx(P, t.wave, k) * SZ_B + max(boff, 0)];
        const int l13 = min(t.lane, 12), l4 = t.lane & 3;
        sbv = gm(P.b)[abidx(P, t.wave, k) * SZ_V13 + t.q * 13 + l13];
        sx = v13(P.xit, t, N + 1, k, l13);
        sy = gm(P.yref)[(t.wave * N + k) * SZ_Y + t.q * 17 + l13];
        su = gm(P.uit)[((size_t)t.inst * N + k) * 4 + l4];
        syu = gm(P.yref)[(t.wave * N + k) * SZ_Y + t.q * 17 + 13 + l4];
    };
    load_stage(k0);
    for (int i = 0; i < m; i++) {
        const int k = k0 + i;
        __syncthreads();
        // stage data -> LDS (dense A, B; b; q = Q (xbar - yref), r = R (ubar - yref_u)); the loads of stage
        // i + 1 were issued before the arithmetic of stage i (clamped addresses, no branches around loads)
        if (aoff[0] >= 0) Am[adst[0]] = sa0;
        if (aoff[1] >= 0) Am[adst[1]] = sa1;
        if (boff >= 0) Bm[t.lane] = sb;
        if (t.lane < 13) {
            bv[t.lane] = sbv;
            qv[t.lane] = wq[t.lane] * (sx - sy);
        }
        if (t.lane < 4) rv[t.lane] = wr[t.lane] * (su - syu);
        if (i + 1 < m) load_stage(k + 1);
        __syncthreads();
        // columns G_i can be non-zero in: inputs of stages < i, dx, 1  (na of them)
        const int na = 4 * i + 14;
        // Gq = diag(Q) G, with the stage's gradient folded into its affine column: then
        // H[r][c] += sum_l Gq[l][r] G[l][c] covers the quadratic AND the linear term (aff is the last index)
        for (int e = t.lane; e < 13 * na; e += LPI) {
            const int l = e / na, ca = e - l * na;
            const int c = ca < 4 * i ? ca : mu + (ca - 4 * i);
            Gq[l * W + c] = wq[l] * G[l * W + c] + (c == aff ? qv[l] : 0.0);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = t.lane + LPI * u;
            if (e < (na * (na + 1)) / 2) {
                const int ra = pk[u] & 255, ca = pk[u] >> 8;
                const int r = ra < 4 * i ? ra : mu + (ra - 4 * i);
                const int c = ca < 4 * i ? ca : mu + (ca - 4 * i);
                const double* gq = Gq + r;
                const double* gg = G + c;
                double acc = 0.0;
#pragma unroll
                for (int l = 0; l < 13; l++) acc += gq[l * W] * gg[l * W];
                H[tri(r, c)] += acc;
            }
        }
        if (t.lane < 4) {
            H[tri(4 * i + t.lane, 4 * i + t.lane)] += wr[t.lane];
            H[tri(aff, 4 * i + t.lane)] += rv[t.lane];
        }
        __syncthreads();
        // G <- A G + [B at the inputs of stage i] + [b at 1]; a lane owns whole columns
        const int nb = 4 * (i + 1) + 14;
        for (int ca = t.lane; ca < nb; ca += LPI) {
            const int c = ca < 4 * (i + 1) ? ca : mu + (ca - 4 * (i + 1));
            double gc[13], gn[13];
#pragma unroll
            for (int l = 0; l < 13; l++) gc[l] = G[l * W + c];
            // A is block upper triangular in the internal order p | v | q | w with an identity p-block
            // (cfnmpc_ws.hpp): row r only meets the columns from its own block on
#pragma unroll
            for (int r = 0; r < 13; r++) {
                const int l0 = r < 6 ? 3 : (r < 10 ? 6 : 10);
                double acc = r < 3 ? gc[r] : 0.0;
#pragma unroll
                for (int l = l0; l < 13; l++) acc += Am[r * 13 + l] * gc[l];
                gn[r] = acc;
            }
            if (c >= 4 * i && c < 4 * i + 4) {
#pragma unroll
                for (int r = 0; r < 13; r++) gn[r] += Bm[r * 4 + (c - 4 * i)];
            }
            if (c == aff) {
#pragma unroll
                for (int r = 0; r < 13; r++) gn[r] += bv[r];
            }
#pragma unroll
            for (int r = 0; r < 13; r++) G[r * W + c] = gn[r];
        }
    }
    __syncthreads();
    gdouble* cb = cb_ptr(P, t, j);
    for (int e = t.lane; e < cond_tri(w); e += LPI) cb[e] = H[e];
    for (int e = t.lane; e < 13 * w; e += LPI) {
        const int r = e / w, c = e - r * w;
        cb[cond_tri(w) + e] = G[r * W + c];
    }
}

// ---------------------------------------------------------------------------------------------
// condensed Riccati stage
// ---------------------------------------------------------------------------------------------
template <int MMAX, int LPI>
struct CfLds {   // LDS carve-up of one group in k_cfactor / k_cipm
    static constexpr int W = cond_w(MMAX);
    static constexpr int U = (cond_tri(W) + LPI - 1) / LPI;   // triangle entries per lane
    static constexpr int SZ = (cond_tri(W) + 13 * W + 14 * 14 + 14 * W + 4 * MMAX + 13 + 17 + 1) & ~1;
    double *H, *Dm, *Pt, *Y, *dinv, *xs, *wv;
    int pk[U];   // (row | column << 8) of the lane's entries lane, lane + LPI, ... of a row-major lower triangle
    __device__ __forceinline__ explicit CfLds(double* base, int lane) {
        H = base; Dm = H + cond_tri(W); Pt = Dm + 13 * W; Y = Pt + 196; dinv = Y + 14 * W; xs = dinv + 4 * MMAX; wv = xs + 13;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = lane + LPI * u, r = tri_row(e);
            pk[u] = r | ((e - (r * (r + 1)) / 2) << 8);
        }
    }
};

__device__ __forceinline__ double p0_of(double piv, double inv) { return piv * inv; }   // sqrt(piv) from its reciprocal root

// One block of the backward recursion.  ABSOLUTE: start solve with the QP's own affine terms;
// otherwise the homogeneous Newton system of the interior point: the condensed Hessian gets
// (Rh - R) on the diagonal of its input block and P.g as its input gradient, all other affine terms
// are zero.  On entry Pt = augmented cost-to-go behind the block, on exit in front of it.
template <int MMAX, int LPI, bool ABSOLUTE>
__device__ __forceinline__ bool cfactor_block(const Params& P, const Grp<LPI>& t, const int j, CfLds<MMAX, LPI>& L) {
    const int N = P.N;
    const int m = cond_len(P, j), k0 = cond_start(P, j);
    const int mu = 4 * m, w = mu + 14, aff = w - 1;
    double *H = L.H, *Dm = L.Dm, *Pt = L.Pt, *Y = L.Y;
    const gdouble* cb = cb_ptr(P, t, j);
    const size_t eb = ((size_t)t.inst * N + k0) * 4;   // element-wise arrays of this block's inputs
    constexpr int W = CfLds<MMAX, LPI>::W;
    __syncthreads();
    {   // the block's H and D: ALL loads first (one HBM round trip instead of one per loop iteration)
        double hr[CfLds<MMAX, LPI>::U];
#pragma unroll
        for (int u = 0; u < CfLds<MMAX, LPI>::U; u++) hr[u] = cb[min(t.lane + LPI * u, cond_tri(w) - 1)];
        for (int c = t.lane; c < w; c += LPI) {
            double dr[13];
#pragma unroll
            for (int l = 0; l < 13; l++) dr[l] = cb[cond_tri(w) + l * w + c];
#pragma unroll
            for (int l = 0; l < 13; l++) Dm[l * W + c] = dr[l];
        }
#pragma unroll
        for (int u = 0; u < CfLds<MMAX, LPI>::U; u++) if (t.lane + LPI * u < cond_tri(w)) H[t.lane + LPI * u] = hr[u];
    }
    __syncthreads();
    if (!ABSOLUTE) {
        for (int c = t.lane; c < w; c += LPI) H[tri(aff, c)] = c < mu ? gm(P.g)[eb + c] : 0.0;
        for (int c = t.lane; c < mu; c += LPI) H[tri(c, c)] += gm(P.Rh)[eb + c] - L.wv[13 + (c & 3)];
        if (t.lane < 13) Dm[t.lane * W + aff] = 0.0;
        __syncthreads();
    }
    // Y = P~ E,  E = [D; e_aff]  (14 x w): a lane owns a column of D
    for (int c = t.lane; c < w; c += LPI) {
        double dc[13];
#pragma unroll
        for (int l = 0; l < 13; l++) dc[l] = Dm[l * W + c];
#pragma unroll
        for (int i = 0; i < 14; i++) {
            double acc = c == aff ? Pt[i * 14 + 13] : 0.0;
#pragma unroll
            for (int l = 0; l < 13; l++) acc += Pt[i * 14 + l] * dc[l];
            Y[i * W + c] = acc;
        }
    }
    __syncthreads();
    // H~ = H + E'Y  (lower triangle, entries dealt out lane by lane)
#pragma unroll
    for (int u = 0; u < CfLds<MMAX, LPI>::U; u++) {
        const int e = t.lane + LPI * u;
        if (e < cond_tri(w)) {
            const int r = L.pk[u] & 255, c = L.pk[u] >> 8;
            const double* dr = Dm + r;
            const double* yc = Y + c;
            double acc = r == aff ? yc[13 * W] : 0.0;
#pragma unroll
            for (int i = 0; i < 13; i++) acc += dr[i * W] * yc[i * W];
            H[e] += acc;
        }
    }
    __syncthreads();
    // right-looking Cholesky of the input block (columns 0 .. mu-1) in PANELS of four columns (mu = 4 m);
    // the trailing rows / columns carry L_xu and the Schur complement.  Per panel: (1) every lane owns
    // one row below / inside the panel: it factors the 4 x 4 diagonal block redundantly (10 broadcast
    // reads, four reciprocal square roots) and solves its row against it, the finished panel also goes
    // to `Lp` as [row][4]; (2) rank-4 update of the trailing triangle, entries dealt out lane by lane:
    // entry e of that triangle is (rr, cc) from the lane's table whatever the panel, at
    // H[e + c1 rr + tri(c1, c1)] with c1 the first column behind the panel.
    bool ok = true;
    double* Lp = Y;
    for (int c0 = 0; c0 < mu; c0 += 4) {
        double dd[10];
#pragma unroll
        for (int e = 0; e < 10; e++) dd[e] = H[tri(c0, c0) + (e < 1 ? 0 : (e < 3 ? c0 + e : (e < 6 ? 2 * c0 + e : 3 * c0 + e)))];
        // dd = [d00 | d10 d11 | d20 d21 d22 | d30 d31 d32 d33]  (rows c0 .. c0+3 of the packed triangle)
        ok = ok && (dd[0] > 0.0);
        const double i0 = rsqrt_nr(dd[0]);
        const double l10 = dd[1] * i0, l20 = dd[3] * i0, l30 = dd[6] * i0;
        const double p1 = dd[2] - l10 * l10;
        ok = ok && (p1 > 0.0);
        const double i1 = rsqrt_nr(p1);
        const double l21 = (dd[4] - l20 * l10) * i1, l31 = (dd[7] - l30 * l10) * i1;
        const double p2 = dd[5] - l20 * l20 - l21 * l21;
        ok = ok && (p2 > 0.0);
        const double i2 = rsqrt_nr(p2);
        const double l32 = (dd[8] - l30 * l20 - l31 * l21) * i2;
        const double p3 = dd[9] - l30 * l30 - l31 * l31 - l32 * l32;
        ok = ok && (p3 > 0.0);
        const double i3 = rsqrt_nr(p3);
        __syncthreads();   // everybody has read the diagonal block
        for (int r = c0 + t.lane; r < w; r += LPI) {
            double* hr = H + tri(r, c0);
            double x0, x1, x2, x3;
            if (r >= c0 + 4) {
                x0 = hr[0] * i0;
                x1 = (hr[1] - x0 * l10) * i1;
                x2 = (hr[2] - x0 * l20 - x1 * l21) * i2;
                x3 = (hr[3] - x0 * l30 - x1 * l31 - x2 * l32) * i3;
                hr[0] = x0; hr[1] = x1; hr[2] = x2; hr[3] = x3;
            } else {   // rows of the diagonal block itself: L_pp (the packed row holds r - c0 + 1 entries)
                const int j = r - c0;
                x0 = j == 0 ? p0_of(dd[0], i0) : (j == 1 ? l10 : (j == 2 ? l20 : l30));
                x1 = j == 1 ? p0_of(p1, i1) : (j == 2 ? l21 : (j == 3 ? l31 : 0.0));
                x2 = j == 2 ? p0_of(p2, i2) : (j == 3 ? l32 : 0.0);
                x3 = j == 3 ? p0_of(p3, i3) : 0.0;
                hr[0] = x0;
                if (j >= 1) hr[1] = x1;
                if (j >= 2) hr[2] = x2;
                if (j >= 3) hr[3] = x3;
            }
            Lp[4 * r + 0] = x0; Lp[4 * r + 1] = x1; Lp[4 * r + 2] = x2; Lp[4 * r + 3] = x3;
        }
        if (t.lane == 0) { L.dinv[c0] = i0; L.dinv[c0 + 1] = i1; L.dinv[c0 + 2] = i2; L.dinv[c0 + 3] = i3; }
        __syncthreads();
        const int c1 = c0 + 4, sdim = w - c1, T = (sdim * (sdim + 1)) / 2;
        double* Hk = H + tri(c1, c1);
        const double* Lk = Lp + 4 * c1;
#pragma unroll
        for (int u = 0; u < CfLds<MMAX, LPI>::U; u++) {
            const int e = t.lane + LPI * u;
            if (e < T) {
                const int rr = L.pk[u] & 255, cc = L.pk[u] >> 8;
                const double* lr = Lk + 4 * rr;
                const double* lc = Lk + 4 * cc;
                Hk[e + c1 * rr] -= lr[0] * lc[0] + lr[1] * lc[1] + lr[2] * lc[2] + lr[3] * lc[3];
            }
        }
        __syncthreads();
    }
    // [K | d]' = L_xu L_uu^-1 for the 14 trailing rows, right-looking from the last PANEL: (A) lane i < 14
    // solves its row against the panel's 4 x 4 triangle (finished entries also to `Xp` as [row][4]),
    // (B) their contribution leaves the columns before the panel: lane = (row i, column group), no
    // index decoding
    double* Xp = Y + 4 * W;
    const int li = t.lane & 15, lg = t.lane >> 4;
    for (int c0 = mu - 4; c0 >= 0; c0 -= 4) {
        if (t.lane < 14) {
            double* hr = H + tri(mu + t.lane, c0);
            const double l10 = H[tri(c0 + 1, c0)], l20 = H[tri(c0 + 2, c0)], l21 = H[tri(c0 + 2, c0 + 1)];
            const double l30 = H[tri(c0 + 3, c0)], l31 = H[tri(c0 + 3, c0 + 1)], l32 = H[tri(c0 + 3, c0 + 2)];
            const double x3 = hr[3] * L.dinv[c0 + 3];
            const double x2 = (hr[2] - x3 * l32) * L.dinv[c0 + 2];
            const double x1 = (hr[1] - x3 * l31 - x2 * l21) * L.dinv[c0 + 1];
            const double x0 = (hr[0] - x3 * l30 - x2 * l20 - x1 * l10) * L.dinv[c0];
            hr[0] = x0; hr[1] = x1; hr[2] = x2; hr[3] = x3;
            Xp[4 * t.lane + 0] = x0; Xp[4 * t.lane + 1] = x1; Xp[4 * t.lane + 2] = x2; Xp[4 * t.lane + 3] = x3;
        }
        __syncthreads();
        if (li < 14) {
            const double x0 = Xp[4 * li], x1 = Xp[4 * li + 1], x2 = Xp[4 * li + 2], x3 = Xp[4 * li + 3];
            double* hr = H + tri(mu + li, 0);
            const double* l0 = H + tri(c0, 0);
            const double* l1 = H + tri(c0 + 1, 0);
            const double* l2 = H + tri(c0 + 2, 0);
            const double* l3 = H + tri(c0 + 3, 0);
            for (int c2 = lg; c2 < c0; c2 += LPI / 16) hr[c2] -= x0 * l0[c2] + x1 * l1[c2] + x2 * l2[c2] + x3 * l3[c2];
        }
        __syncthreads();
    }
    // gains per ORIGINAL stage in the layout of the uncondensed path: lane a of stage k holds K[a][0..12]
    for (int e = t.lane; e < 13 * mu; e += LPI) {
        const int c = e / 13, l = e - c * 13;
        const int k = k0 + (c >> 2), a = c & 3;
        gm(P.KR)[(t.wave * N + k) * SZ_K + (l * 4 + t.q) * 4 + a] = H[tri(mu + l, c)];
    }
    for (int c = t.lane; c < mu; c += LPI) gm(P.d)[eb + c] = H[tri(mu + 13, c)];
    // cost-to-go in front of the block
    for (int e = t.lane; e < 196; e += LPI) {
        const int i = e / 14, l = e - i * 14;
        Pt[e] = (i == 13 && l == 13) ? 0.0 : H[trs(mu + i, mu + l)];
    }
    return ok;
}

template <int MMAX, int LPI, bool ABSOLUTE>
__device__ __forceinline__ bool csweep_factor(const Params& P, const Grp<LPI>& t, CfLds<MMAX, LPI>& L) {
    const int N = P.N;
    __syncthreads();
    for (int e = t.lane; e < 196; e += LPI) L.Pt[e] = 0.0;
    __syncthreads();
    if (t.lane < 13) {
        const double wn = P.WN[ext_of(t.lane)];
        L.Pt[t.lane * 14 + t.lane] = wn;
        if (ABSOLUTE) {
            const double xN = v13(P.xit, t, N + 1, N, t.lane);
            const double yN = gm(P.yref_e)[t.wave * SZ_V13 + t.q * 13 + t.lane];
            const double qn = wn * (xN - yN);
            L.Pt[t.lane * 14 + 13] = qn;
            L.Pt[13 * 14 + t.lane] = qn;
        }
    }
    bool ok = true;
    for (int j = P.cond_N2 - 1; j >= 0; j--) ok = cfactor_block<MMAX, LPI, ABSOLUTE>(P, t, j, L) && ok;
    return ok;
}

template <int MMAX, int LPI>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MMAX <= 5 ? 3 : 2, MMAX <= 5 ? 3 : 2))) void k_cfactor(Params P) {
    constexpr int IPW = 64 / LPI;
    __shared__ double lds[IPW * CfLds<MMAX, LPI>::SZ];
    const Grp<LPI> t = grp_id<LPI>(P, blockIdx.x * IPW);
    CfLds<MMAX, LPI> L(lds + t.g * CfLds<MMAX, LPI>::SZ, t.lane);
    if (t.lane < 17) L.wv[t.lane] = P.W[t.lane < 13 ? ext_of(t.lane) : t.lane];
    bool ok = csweep_factor<MMAX, LPI, true>(P, t, L);
    ok = grp_min<LPI>(ok ? 1.0 : 0.0) > 0.0;
    if (t.lane == 0 && t.valid) gm(P.status)[t.inst] = ok ? 0 : 4;
}

// forward sweep of a homogeneous (delta) solve on the condensed blocks: dU_j = -K dx - d at the
// block start, dx+ = Abar dx + Bbar dU.  Writes the input step (all N stages) to `out`.
template <int MMAX, int LPI>
__device__ __forceinline__ void csweep_forward_delta(const Params& P, const Grp<LPI>& t, CfLds<MMAX, LPI>& L, double* out) {
    // Per block: dU = -(d + K dx), dx+ = D [dU; dx].  K (4m x 13, just written by the factorisation), d and
    // the block's D come from global memory: all lanes fetch them in ONE batch into registers -- for
    // block j + 1 before the arithmetic of block j -- and hand them over through LDS (Dm and Y are free
    // between factorisations), so a block costs no exposed round trip.
    constexpr int W = CfLds<MMAX, LPI>::W;
    constexpr int ND = (13 * W + LPI - 1) / LPI, NK = (13 * 4 * MMAX + LPI - 1) / LPI;
    static_assert(LPI >= 4 * MMAX, "one input of a block per lane (dd)");
    const int N = P.N;
    double* xs = L.xs;              // dx at the block start
    double* Kl = L.Y;               // [c][13]
    double* U = L.Y + 13 * 4 * MMAX;   // dU of the block
    double* Dm = L.Dm;              // [13][W]
    double dr[ND], kr[NK], dd;
    auto fetch = [&](const int j) {
        const int m = cond_len(P, j), k0 = cond_start(P, j);
        const int mu = 4 * m, w = mu + 14;
        const gdouble* cbD = cb_ptr(P, t, j) + cond_tri(w);
#pragma unroll
        for (int u = 0; u < ND; u++) {
            const int e = t.lane + LPI * u, l = min(e / W, 12), c = min(e - (e / W) * W, w - 1);
            dr[u] = cbD[l * w + c];
        }
#pragma unroll
        for (int u = 0; u < NK; u++) {
            const int e = t.lane + LPI * u, c = min(e / 13, mu - 1), l = e - (e / 13) * 13;
            kr[u] = gm(P.KR)[(t.wave * N + k0 + (c >> 2)) * SZ_K + (l * 4 + t.q) * 4 + (c & 3)];
        }
        dd = gm(P.d)[((size_t)t.inst * N + k0) * 4 + min(t.lane, mu - 1)];
    };
    __syncthreads();
    if (t.lane < 13) xs[t.lane] = 0.0;
    fetch(0);
    for (int j = 0; j < P.cond_N2; j++) {
        const int m = cond_len(P, j), k0 = cond_start(P, j);
        const int mu = 4 * m, w = mu + 14;
        const size_t eb = ((size_t)t.inst * N + k0) * 4;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < ND; u++) {
            const int e = t.lane + LPI * u, l = e / W, c = e - l * W;
            if (l < 13 && c < w) Dm[e] = dr[u];
        }
#pragma unroll
        for (int u = 0; u < NK; u++) {
            const int e = t.lane + LPI * u;
            if (e < 13 * mu) Kl[e] = kr[u];
        }
        const double dj = dd;
        if (j + 1 < P.cond_N2) fetch(j + 1);
        __syncthreads();
        for (int c = t.lane; c < mu; c += LPI) {
            double acc = dj;
#pragma unroll
            for (int l = 0; l < 13; l++) acc += Kl[c * 13 + l] * xs[l];
            U[c] = -acc;
            gm(out)[eb + c] = -acc;
        }
        __syncthreads();
        double xn = 0.0;
        if (t.lane < 13) {
            const double* Dr = Dm + t.lane * W;
            for (int c = 0; c < mu; c++) xn += Dr[c] * U[c];
#pragma unroll
            for (int l = 0; l < 13; l++) xn += Dr[mu + l] * xs[l];
        }
        __syncthreads();
        if (t.lane < 13) xs[t.lane] = xn;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// interior point on the condensed QP (instances whose unconstrained minimiser leaves the box)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double ratio(double z, double dz, double a) {
    const double tt = -z * rcp_nr(dz);
    return (dz < 0.0 && tt < a) ? tt : a;
}

template <int MMAX, int LPI>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_cipm(Params P) {
    constexpr int IPW = 64 / LPI;
    __shared__ double lds[IPW * CfLds<MMAX, LPI>::SZ];
    const Grp<LPI> t = grp_id<LPI>(P, blockIdx.x * IPW);
    const double viol = t.valid ? gm(P.viol)[t.inst] : 0.0;
    const bool infeasible = t.valid && viol > 0.0 && gm(P.status)[t.inst] == 0;
    if (!__any(infeasible)) return;
    CfLds<MMAX, LPI> L(lds + t.g * CfLds<MMAX, LPI>::SZ, t.lane);
    if (t.lane < 17) L.wv[t.lane] = P.W[t.lane < 13 ? ext_of(t.lane) : t.lane];
    __syncthreads();
    const int N = P.N, n = 4 * N;
    const size_t eb = (size_t)t.inst * N * 4;
    gdouble *v = gm(P.v) + eb, *tl = gm(P.tl) + eb, *tu = gm(P.tu) + eb, *ll = gm(P.ll) + eb, *lu = gm(P.lu) + eb,
            *rg = gm(P.rg) + eb, *Rh = gm(P.Rh) + eb, *g = gm(P.g) + eb, *dva = gm(P.dva) + eb, *dvc = gm(P.dvc) + eb;
    const gdouble* uit = gm(P.uit) + eb;
    // ---- start: slacks / multipliers shifted positive (delta form around the unconstrained minimiser)
    double mu = 0.0, res = 0.0;
    int iters = 0, status = infeasible ? 2 : 0;
    bool act = infeasible;
    {
        const double mu0 = fmax(P.mu0_scale * viol, P.lam0_min);
        double smu = 0.0, sres = 0.0;
        for (int e = t.lane; e < n; e += LPI) {
            const double wu = L.wv[13 + (e & 3)];
            if (!infeasible) { Rh[e] = wu; g[e] = 0.0; continue; }   // benign zero solve for the wave-mates
            const double vv = v[e], uk = uit[e];
            const double lb = P.u_min - uk, ub = P.u_max - uk;
            const double tle = fmax(vv - lb, P.thr0), tue = fmax(ub - vv, P.thr0);
            const double itl = rcp_nr(tle), itu = rcp_nr(tue);
            const double lle = mu0 * itl, lue = mu0 * itu, rge = -lle + lue;
            tl[e] = tle; tu[e] = tue; ll[e] = lle; lu[e] = lue; rg[e] = rge;
            const double rl = vv - lb - tle, ru = ub - vv - tue;
            const double Dl = lle * itl, Du = lue * itu;
            Rh[e] = wu + Dl + Du;
            g[e] = rge + lle + Dl * rl - lue - Du * ru;
            smu += lle * tle + lue * tue;
            sres = fmax(sres, fmax(fmax(lle * tle, lue * tue), fmax(fabs(rge), fmax(fabs(rl), fabs(ru)))));
        }
        mu = grp_sum<LPI>(smu) / (2.0 * n);
        res = grp_max<LPI>(sres);
    }
    while (__any(act)) {
        if (act) {
            if (!(res == res)) { status = 4; act = false; }
            else if (res <= P.tol) { status = 0; act = false; }
            else if (iters >= P.max_iter) { status = 2; act = false; }
        }
        if (!__any(act)) break;
        if (act) iters++;
        // predictor
        bool fok = csweep_factor<MMAX, LPI, false>(P, t, L);
        csweep_forward_delta<MMAX, LPI>(P, t, L, P.dva);
        __syncthreads();
        double smu;
        {
            double a = 1.0;
            for (int e = t.lane; e < n && act; e += LPI) {
                const double uk = uit[e], lb = P.u_min - uk, ub = P.u_max - uk;
                const double rl = v[e] - lb - tl[e], ru = ub - v[e] - tu[e];
                const double dtl = dva[e] + rl, dtu = -dva[e] + ru;
                const double dll = -ll[e] - (ll[e] * rcp_nr(tl[e])) * dtl, dlu = -lu[e] - (lu[e] * rcp_nr(tu[e])) * dtu;
                a = ratio(tl[e], dtl, a); a = ratio(tu[e], dtu, a); a = ratio(ll[e], dll, a); a = ratio(lu[e], dlu, a);
            }
            a = grp_min<LPI>(a);
            double mu_aff = 0.0;
            for (int e = t.lane; e < n && act; e += LPI) {
                const double uk = uit[e], lb = P.u_min - uk, ub = P.u_max - uk;
                const double rl = v[e] - lb - tl[e], ru = ub - v[e] - tu[e];
                const double dtl = dva[e] + rl, dtu = -dva[e] + ru;
                const double dll = -ll[e] - (ll[e] * rcp_nr(tl[e])) * dtl, dlu = -lu[e] - (lu[e] * rcp_nr(tu[e])) * dtu;
                mu_aff += (ll[e] + a * dll) * (tl[e] + a * dtl) + (lu[e] + a * dlu) * (tu[e] + a * dtu);
            }
            mu_aff = grp_sum<LPI>(mu_aff) / (2.0 * n);
            const double sr = mu_aff * rcp_nr(mu);
            smu = sr * sr * sr * mu;
            for (int e = t.lane; e < n && act; e += LPI) {
                const double uk = uit[e], lb = P.u_min - uk, ub = P.u_max - uk;
                const double rl = v[e] - lb - tl[e], ru = ub - v[e] - tu[e];
                const double dtl = dva[e] + rl, dtu = -dva[e] + ru;
                const double itl = rcp_nr(tl[e]), itu = rcp_nr(tu[e]);
                const double dll = -ll[e] - (ll[e] * itl) * dtl, dlu = -lu[e] - (lu[e] * itu) * dtu;
                g[e] = (dll * dtl - smu) * itl - (dlu * dtu - smu) * itu;
            }
        }
        __syncthreads();
        // corrector: same matrix, new gradient
        fok = csweep_factor<MMAX, LPI, false>(P, t, L) && fok;
        csweep_forward_delta<MMAX, LPI>(P, t, L, P.dvc);
        __syncthreads();
        {
            double a = 1.0;
            for (int e = t.lane; e < n && act; e += LPI) {
                const double uk = uit[e], lb = P.u_min - uk, ub = P.u_max - uk;
                const double dv = dva[e] + dvc[e];
                const double rl = v[e] - lb - tl[e], ru = ub - v[e] - tu[e];
                const double dtla = dva[e] + rl, dtua = -dva[e] + ru;
                const double itl = rcp_nr(tl[e]), itu = rcp_nr(tu[e]);
                const double Dl = ll[e] * itl, Du = lu[e] * itu;
                const double cl = (-ll[e] - Dl * dtla) * dtla, cu = (-lu[e] - Du * dtua) * dtua;
                const double dtl = dv + rl, dtu = -dv + ru;
                const double dll = (smu - cl) * itl - ll[e] - Dl * dtl, dlu = (smu - cu) * itu - lu[e] - Du * dtu;
                a = ratio(tl[e], dtl, a); a = ratio(tu[e], dtu, a); a = ratio(ll[e], dll, a); a = ratio(lu[e], dlu, a);
            }
            a = fmin(1.0, P.tau * grp_min<LPI>(a));
            double smu2 = 0.0, sres = 0.0;
            for (int e = t.lane; e < n && act; e += LPI) {
                const double wu = L.wv[13 + (e & 3)];
                const double uk = uit[e], lb = P.u_min - uk, ub = P.u_max - uk;
                const double dv = dva[e] + dvc[e];
                const double rl = v[e] - lb - tl[e], ru = ub - v[e] - tu[e];
                const double dtla = dva[e] + rl, dtua = -dva[e] + ru;
                const double itl = rcp_nr(tl[e]), itu = rcp_nr(tu[e]);
                const double Dl = ll[e] * itl, Du = lu[e] * itu;
                const double cl = (-ll[e] - Dl * dtla) * dtla, cu = (-lu[e] - Du * dtua) * dtua;
                const double dtl = dv + rl, dtu = -dv + ru;
                const double dll = (smu - cl) * itl - ll[e] - Dl * dtl, dlu = (smu - cu) * itu - lu[e] - Du * dtu;
                const double vn = v[e] + a * dv, tln = tl[e] + a * dtl, tun = tu[e] + a * dtu;
                const double lln = ll[e] + a * dll, lun = lu[e] + a * dlu, rgn = rg[e] * (1.0 - a);
                const double rln = vn - lb - tln, run = ub - vn - tun;
                const double Dln = lln * rcp_nr(tln), Dun = lun * rcp_nr(tun);
                v[e] = vn; tl[e] = tln; tu[e] = tun; ll[e] = lln; lu[e] = lun; rg[e] = rgn;
                Rh[e] = wu + Dln + Dun;
                g[e] = rgn + lln + Dln * rln - lun - Dun * run;
                smu2 += lln * tln + lun * tun;
                sres = fmax(sres, fmax(fmax(lln * tln, lun * tun), fmax(fabs(rgn), fmax(fabs(rln), fabs(run)))));
            }
            smu2 = grp_sum<LPI>(smu2) / (2.0 * n);
            sres = grp_max<LPI>(sres);
            const bool fok_g = grp_min<LPI>(fok ? 1.0 : 0.0) > 0.0;
            if (act) {
                mu = smu2;
                res = fok_g ? sres : nan("");
            }
        }
        __syncthreads();
    }
    // ---- expand: roll the final inputs through the stored stage dynamics; new iterate = old + step
    //      (a failed QP keeps the old iterate)
    {   // (wave-uniform control flow: the wave-mates walk along with their stores masked)
        const bool take = infeasible && status != 4;
        double* xs = L.xs;
        __syncthreads();
        if (t.lane < 13) xs[t.lane] = v13(P.x0, t, 1, 0, t.lane) - v13(P.xit, t, N + 1, 0, t.lane);
        __syncthreads();
        for (int k = 0; k < N; k++) {
            double du[4];
            for (int a = 0; a < 4; a++) du[a] = take ? v[k * 4 + a] : 0.0;
            if (t.lane < 4 && infeasible) gm(P.uitn)[eb + k * 4 + t.lane] = uit[k * 4 + t.lane] + du[t.lane];
            double xn = 0.0;
            if (t.lane < 13) {
                const double xb = v13(P.xit, t, N + 1, k, t.lane);
                if (infeasible) gm(P.xitn)[(t.wave * (N + 1) + k) * SZ_V13 + t.q * 13 + t.lane] = xb + (take ? xs[t.lane] : 0.0);
                xn = gm(P.b)[abidx(P, t.wave, k) * SZ_V13 + t.q * 13 + t.lane];
                for (int c = 0; c < 13; c++) xn += a_elem(P, t, k, t.lane, c) * xs[c];
                for (int a = 0; a < 4; a++) xn += b_elem(P, t, k, t.lane, a) * du[a];
            }
            __syncthreads();
            if (t.lane < 13) xs[t.lane] = xn;
            __syncthreads();
        }
        if (t.lane < 13 && infeasible) {
            const double xb = v13(P.xit, t, N + 1, N, t.lane);
            gm(P.xitn)[(t.wave * (N + 1) + N) * SZ_V13 + t.q * 13 + t.lane] = xb + (take ? xs[t.lane] : 0.0);
        }
        if (t.lane == 0 && infeasible) {
            gm(P.status)[t.inst] = status;
            gm(P.iters)[t.inst] = iters;
            gm(P.res)[t.inst] = res;
            gm(P.head)[t.inst] = N;
        }
    }
}

template <int MMAX, int LPI>
void launch_pcond_t(const Params& P, hipStream_t st) {
    constexpr int IPW = 64 / LPI;
    hipLaunchKernelGGL((k_pcond<MMAX, LPI>), dim3((P.B + IPW - 1) / IPW, P.cond_N2), dim3(64), 0, st, P);
}
template <int MMAX, int LPI>
void launch_cfactor_t(const Params& P, hipStream_t st) {
    constexpr int IPW = 64 / LPI;
    hipLaunchKernelGGL((k_cfactor<MMAX, LPI>), dim3((P.B + IPW - 1) / IPW), dim3(64), 0, st, P);
}
template <int MMAX, int LPI>
void launch_cipm_t(const Params& P, hipStream_t st) {
    constexpr int IPW = 64 / LPI;
    hipLaunchKernelGGL((k_cipm<MMAX, LPI>), dim3((P.B + IPW - 1) / IPW), dim3(64), 0, st, P);
}

}  // namespace

// template instances: block lengths up to 2, 5 and 10 stages; one instance per wavefront (LPI = 64):
// the dense blocks of an instance take 6 - 26 KB of LDS, so a CU holds as many INSTANCES either way,
// and with all 64 lanes on one instance every phase finishes ~4x sooner (measured: 16 lanes per
// instance 3 - 5x slower) and the interior-point kernel only occupies waves that have work
#define CFN_COND_DISPATCH(fn)                              \
    do {                                                   \
        const int mm = cond_mmax(P);                       \
        if (mm <= 2) fn<2, 64>(P, st);                     \
        else if (mm <= 5) fn<5, 64>(P, st);                \
        else fn<10, 64>(P, st);                            \
    } while (0)

void launch_pcond(const Params& P, hipStream_t st) { CFN_COND_DISPATCH(launch_pcond_t); }
void launch_cfactor(const Params& P, hipStream_t st) { CFN_COND_DISPATCH(launch_cfactor_t); }
void launch_cipm(const Params& P, hipStream_t st) { CFN_COND_DISPATCH(launch_cipm_t); }
void launch_qp_cond(const Params& P, hipStream_t st) {
    launch_pcond(P, st);
    launch_cfactor(P, st);
    launch_cforward(P, st);
    launch_cipm(P, st);
}

}  // namespace cfn
