// Fused PointTransformerLayer passes on the matrix cores, for nsample == 16 and C = 64 / 128 / 256 / 512 (levels 2-5: 50k / 12.5k /
// 3.1k / 780 points).  Same math and buffers as fused_layer.hip (which documents the pass structure and cites
// point_transformer_seg.py:45-78); only the mapping differs.
//
// The row-per-lane kernels keep one (point, neighbour) row per lane and stream the weights through SGPRs; at C >= 128 that
// is C * C/8 = 2048..8192 serial FMAs per lane and pass behind scalar loads, and the level-4 layers (1.5 % of the points)
// took 17 % of the step.  Here one wave works on ONE point = 16 neighbour rows at a time and the C x C/8 products run as
// v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate):
//
//   lane l: row = l & 15 (neighbour), kq = l >> 4;  channels of the lane = {16 j + 4 kq + e : e < 4} for j < C/16
//   -> every per-channel quantity (x_k[idx] row gather, x_q row, p_r, BatchNorm coefficients) is a float4 per (lane, j):
//      64 contiguous bytes per row and instruction, per-channel constants from LDS (block-wide copy, broadcast reads)
//   -> h^T = Ww1 . relu(BN1(r))^T : A = Ww1 (16 hidden units x 4 channels), B = activations^T; the D fragment
//      (hidden unit 4 (l >> 4) + reg, row l & 15) is again "4 consecutive values per lane": float4 stores of h.
//   Column statistics (sums over rows) are per-lane accumulators reduced over the 16 row-lanes once per kernel.
#include "fused_layer_mfma.h"

namespace flm {

bool supported(int nsample, int c) {
    return nsample == 16 && (c == 64 || c == 128 || c == 256 || c == 512) && getenv("PDFOPS_PT_NO_MFMA") == nullptr;
}

// ------------------------------------------------------------------------------------------------ P2: stats of r (slabs)
// partial row per block: [sum r (C) | sum r^2 (C)]; slab y writes channels [64 y, 64 y + 64)
template <int C>
__global__ __launch_bounds__(64 * WPB) void k_p2(LayerArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cst = lds;
    stage_consts<C>(cst, A, false);
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const int c0 = 64 * blockIdx.y;
    GeoW G = geo_weights(A);
    if (A.mom) {   // BNp from the geometry moments (fused_layer.h); block (0, 0) stores the coefficients for the later passes
        const fl::BnP B = fl::bnp_of(A, (long)A.N * 16, blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0);
#pragma unroll
        for (int a = 0; a < 3; ++a) { G.sp[a] = B.sp[a]; G.tp[a] = B.tp[a]; }
    }
    f32x4 s[4], ss[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) { s[jj] = zero4(); ss[jj] = zero4(); }
    fl::PointWalk pw(A, threadIdx.x >> 6);
    int nb_next = pw.valid() ? A.idx[pw.point() * 16 + row] : -1;
    for (; pw.valid(); pw.step()) {
        const long i = pw.point();
        const int nb = nb_next;
        const size_t nbc = (size_t)max(nb, 0);
        nb_next = A.idx[(pw.has_next() ? pw.next_point() : i) * 16 + row];   // next trip's index: in flight during this trip
        float pn[3], pi[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) { pn[b] = A.p[nbc * 3 + b]; pi[b] = A.p[(size_t)i * 3 + b]; }
        f32x4 xk[4], xq[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int c4 = c0 + 16 * jj + 4 * kq;
            xk[jj] = ld4(A.xk + nbc * C + c4); xq[jj] = ld4(A.xq + (size_t)i * C + c4);
        }
        __builtin_amdgcn_sched_barrier(0);
        const Geo R = geo_of(G, nb, pn, pi);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const f32x4 r = (sel4(nb >= 0, xk[jj]) - xq[jj]) + pos4(cst, C, 4 * (4 * (int)blockIdx.y + jj) + kq, R.t1n);
            s[jj] += r;
            ss[jj] += r * r;
        }
    }
    block_row(lds, 128, [&](RowAcc o) {   // this slab's 64 + 64 columns
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = s[jj][e], b = ss[jj][e];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
                if (row == 0) { o[16 * jj + 4 * kq + e] = a; o[64 + 16 * jj + 4 * kq + e] = b; }
            }
    });
    if (threadIdx.x < 128) (A.partial + (size_t)blockIdx.x * 2 * C)[(threadIdx.x >> 6) * C + c0 + (threadIdx.x & 63)] = lds[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------ P3: h (+ stats of h)
template <int C, bool STATS, bool BF>
__global__ __launch_bounds__(64 * WPB) void k_p3(LayerArgs A) {
    constexpr int CS = C / 8, NG = C / 64, NOB = nob_of(C), CSP = csp_of(C), WS = C + 4;   // WS: padded row stride of the Ww1 copy
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cst = lds, *wl = lds + 6 * C;
    stage_rows<CSP, C, WS>(wl, gp(A.Ww1), C, CS);
    stage_consts<C>(cst, A, true);
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const float *wa = wl + row * WS + 4 * kq;   // A operand (hidden unit ob*16 + (l & 15), k = kq): wa[ob * 16 * WS + 16 j ..+4]
    const GeoW G = geo_weights(A);
    f32x4 b4[NOB], s4[NOB], ss4[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) { b4[ob] = ldu(gp(A.bw1), ob * 16 + 4 * kq, ob * 16 + 4 * kq < CS); s4[ob] = zero4(); ss4[ob] = zero4(); }
    fl::PointWalk pw(A, threadIdx.x >> 6);
    int nb_next = pw.valid() ? A.idx[pw.point() * 16 + row] : -1;
    for (; pw.valid(); pw.step()) {
        const long i = pw.point();
        const int nb = nb_next;
        const size_t nbc = (size_t)max(nb, 0);
        nb_next = A.idx[(pw.has_next() ? pw.next_point() : i) * 16 + row];
        float pn[3], pi[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) { pn[b] = A.p[nbc * 3 + b]; pi[b] = A.p[(size_t)i * 3 + b]; }
        const float *xkr = A.xk + nbc * C + 4 * kq, *xqr = A.xq + (size_t)i * C + 4 * kq;
        f32x4 acc[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) acc[ob] = zero4();
        if constexpr (C == 128 || C == 256) {
            // (measured: at these widths the double-buffered groups cost more in occupancy -- 142 / 162 VGPRs -- than they hide)
            const Geo R = geo_of(G, nb, pn, pi);
#pragma unroll 2
            for (int j = 0; j < C / 16; ++j) {
                const int g = 4 * j + kq;
                const f32x4 r = (sel4(nb >= 0, ld4(xkr + 16 * j)) - ld4(xqr + 16 * j)) + pos4(cst, C, g, R.t1n);
                const f32x4 y = relu4(r * ld4(cst + 4 * C + 4 * g) + ld4(cst + 5 * C + 4 * g));
                f32x4 w[NOB];
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) w[ob] = ld4(wa + ob * 16 * WS + 16 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ob][e], y[e], acc[ob], 0, 0, 0);
            }
        } else {
        f32x4 xk[2][4], xq[2][4];   // double-buffered groups of four 16-channel blocks
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { xk[0][jj] = ld4(xkr + 16 * jj); xq[0][jj] = ld4(xqr + 16 * jj); }
        __builtin_amdgcn_sched_barrier(0);
        const Geo R = geo_of(G, nb, pn, pi);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi + 1 < NG) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { xk[(gi + 1) & 1][jj] = ld4(xkr + 64 * (gi + 1) + 16 * jj); xq[(gi + 1) & 1][jj] = ld4(xqr + 64 * (gi + 1) + 16 * jj); }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 4 * gi + jj, g = 4 * j + kq;
                const f32x4 r = (sel4(nb >= 0, xk[gi & 1][jj]) - xq[gi & 1][jj]) + pos4(cst, C, g, R.t1n);
                const f32x4 y = relu4(r * ld4(cst + 4 * C + 4 * g) + ld4(cst + 5 * C + 4 * g));
                f32x4 w[NOB];
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) w[ob] = ld4(wa + ob * 16 * WS + 16 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ob][e], y[e], acc[ob], 0, 0, 0);
            }
        }
        }
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const f32x4 h = acc[ob] + b4[ob];   // h[row][ob*16 + 4 kq + reg]
            if (ob * 16 + 4 * kq < CS) st_row4<BF>(A.H, ((size_t)i * 16 + row) * CS + ob * 16 + 4 * kq, h);
            if (STATS) { s4[ob] += h; ss4[ob] += h * h; }
        }
    }
    if (STATS) {
        block_row(lds, 2 * CS, [&](RowAcc o) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = s4[ob][r], ss = ss4[ob][r];
#pragma unroll
                    for (int m = 1; m < 16; m <<= 1) { s += __shfl_xor(s, m, 64); ss += __shfl_xor(ss, m, 64); }
                    if (row == 0 && ob * 16 + 4 * kq < CS) { o[ob * 16 + 4 * kq + r] = s; o[CS + ob * 16 + 4 * kq + r] = ss; }
                }
        });
        store_row(lds, 2 * CS, A.partial + (size_t)blockIdx.x * 2 * CS);
    }
}

// ------------------------------------------------------------------------------------------------ P4: aggregation
// out[i][c] = sum_rows (x_v[nb][c] + p_r[c]) * w[row][c mod CS]
// STATS: + one partial row [sum out (C) | sum out^2 (C)] per block in A.partial (the statistics of the Bottleneck's bn2, fl::k_p4).  After
// the sum over the 16 rows every row-lane holds the finished value of its channels; row-lane r accumulates the 16-channel blocks j with
// j mod 16 == r.
template <int C, bool BF, bool STATS>
__global__ __launch_bounds__(64 * WPB) void k_p4(LayerArgs A) {
    constexpr int CS = C / 8, NG = C / 64, NOB = nob_of(C), CSP = csp_of(C), NJ = C / 16, NA = (NJ + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cst = lds, *ucst = lds + 4 * C, *w2 = ucst + 3 * CSP;
    stage_consts<C>(cst, A, false);
    stage_units<C, false>(ucst, A, nullptr);
    stage_w2<C>(w2, A);
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const GeoW G = geo_weights(A);
    f32x4 so[STATS ? NA : 1], sso[STATS ? NA : 1];
    if (STATS) {
#pragma unroll
        for (int a = 0; a < NA; ++a) { so[a] = zero4(); sso[a] = zero4(); }
    }
    fl::PointWalk pw(A, threadIdx.x >> 6);
    int nb_next = pw.valid() ? A.idx[pw.point() * 16 + row] : -1;
    for (; pw.valid(); pw.step()) {
        const long i = pw.point();
        const int nb = nb_next;
        const size_t nbc = (size_t)max(nb, 0);
        nb_next = A.idx[(pw.has_next() ? pw.next_point() : i) * 16 + row];
        float pn[3], pi[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) { pn[b] = A.p[nbc * 3 + b]; pi[b] = A.p[(size_t)i * 3 + b]; }
        f32x4 h[NOB], u[NOB], w[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) h[ob] = ld_row4<BF>(A.H, ((size_t)i * 16 + row) * CS + unit_off<C>(ob, kq));
        const float *xvr = A.xv + nbc * C + 4 * kq;
        f32x4 xv[2][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) xv[0][jj] = ld4(xvr + 16 * jj);
        __builtin_amdgcn_sched_barrier(0);
        const Geo R = geo_of(G, nb, pn, pi);
        attn_weights<C>(ucst, w2, row, kq, h, u, w);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi + 1 < NG) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) xv[(gi + 1) & 1][jj] = ld4(xvr + 64 * (gi + 1) + 16 * jj);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {   // channel 16 j + 4 kq + e -> hidden unit block j % NOB
                const int j = 4 * gi + jj, g = 4 * j + kq;
                const f32x4 v = (sel4(nb >= 0, xv[gi & 1][jj]) + pos4(cst, C, g, R.t1n)) * w[j % NOB];
                f32x4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = sum16(v[e]);
                if (row == 0) st4(A.out + (size_t)i * C + 4 * g, t);
                if (STATS && row == (j & 15)) { so[STATS ? j / 16 : 0] += t; sso[STATS ? j / 16 : 0] += t * t; }
            }
        }
    }
    if (STATS) {
        block_row(lds, 2 * C, [&](RowAcc o) {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int j = 16 * a + row;
                if (j < NJ) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[16 * j + 4 * kq + e] = so[STATS ? a : 0][e]; o[C + 16 * j + 4 * kq + e] = sso[STATS ? a : 0][e]; }
                }
            }
        });
        store_row(lds, 2 * C, A.partial + (size_t)blockIdx.x * 2 * C);
    }
}

// ------------------------------------------------------------------------------------------------ B1
// partial row per block: [sum g_y2 (CS) | sum g_y2*hhat (CS) | g_bw2 (CS) | g_Ww2 (CS*CS)]   (as fl::k_b1)
template <int C, bool BF>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(C == 64 ? 3 : 1, 8))) void k_b1(LayerArgs A) {   // (C = 64: 176 -> 168 registers: 79 -> 73 us with 768 workgroups)
    constexpr int CS = C / 8, NOB = nob_of(C), CSP = csp_of(C), NCHK = C / 64, GS = CSP + 4, WS2 = CSP + 4, W = 3 * CS + CS * CS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cst = lds, *ucst = cst + 4 * C, *w2 = ucst + 7 * CSP;           // Wp2 (3C) | bp2 (C); per-unit constants; padded Ww2
    const int wv = threadIdx.x >> 6;
    float *gz_t = w2 + w2_floats(C) + wv * 16 * GS, *u_t = gz_t + WPB * 16 * GS;
    stage_consts<C>(cst, A, false);
    stage_units<C, false>(ucst, A, nullptr);
    for (int e = threadIdx.x; e < 2 * CSP; e += NT) {   // mean2 | rstd2 (this pass computes the BN2-backward sums: no S yet)
        const int u = min(e % CSP, CS - 1);
        ucst[U_MEAN * CSP + e] = e % CSP < CS ? (e < CSP ? gp(A.mean) : gp(A.rstd))[3 + C + u] : 0.f;
    }
    stage_w2<C>(w2, A);
    for (int e = threadIdx.x; e < 2 * WPB * 16 * GS; e += NT) (w2 + w2_floats(C))[e] = 0.f;   // unit tiles incl. padding columns
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const GeoW G = geo_weights(A);
    f32x4 sg[NOB], sgh[NOB], sgz[NOB], accw[NOB][NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        sg[ob] = zero4(); sgh[ob] = zero4(); sgz[ob] = zero4();
#pragma unroll
        for (int ub = 0; ub < NOB; ++ub) accw[ob][ub] = zero4();
    }
    fl::PointWalk pw(A, wv);
    int nb_next = pw.valid() ? A.idx[pw.point() * 16 + row] : -1;
    for (; pw.valid(); pw.step()) {
        const long i = pw.point();
        const size_t ri = (size_t)i * 16 + row;
        const int nb = nb_next;
        const size_t nbc = (size_t)max(nb, 0);
        nb_next = A.idx[(pw.has_next() ? pw.next_point() : i) * 16 + row];
        float pn[3], pi[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) { pn[b] = A.p[nbc * 3 + b]; pi[b] = A.p[(size_t)i * 3 + b]; }
        f32x4 h[NOB], u[NOB], w[NOB], gw[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) { h[ob] = ld_row4<BF>(A.H, ri * CS + unit_off<C>(ob, kq)); gw[ob] = zero4(); }
        const float *xvr = A.xv + nbc * C + 4 * kq, *gor = A.gout + (size_t)i * C + 4 * kq;
        f32x4 xv[2][4], go[2][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { xv[0][jj] = ld4(xvr + 16 * jj); go[0][jj] = ld4(gor + 16 * jj); }
        __builtin_amdgcn_sched_barrier(0);
        const Geo R = geo_of(G, nb, pn, pi);
        attn_weights<C>(ucst, w2, row, kq, h, u, w);
#pragma unroll
        for (int q = 0; q < NCHK; ++q) {
            if (q + 1 < NCHK) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { xv[(q + 1) & 1][jj] = ld4(xvr + 64 * (q + 1) + 16 * jj); go[(q + 1) & 1][jj] = ld4(gor + 64 * (q + 1) + 16 * jj); }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int g = 4 * (4 * q + jj) + kq;   // channel 16 j + 4 kq + e  ->  hidden unit (c mod CS): block jj % NOB, same (kq, e)
                gw[jj % NOB] += go[q & 1][jj] * (sel4(nb >= 0, xv[q & 1][jj]) + pos4(cst, C, g, R.t1n));
            }
        }
        // softmax weights of the 16 rows: g_xv[nb] = sum over the inverse kNN table of g_out[i] * w  (pdf_seg_sum_weighted); B3 reads them too
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
            if (16 * ob + 4 * kq < CS) st_row4<BF>(A.Wsm, ri * CS + 16 * ob + 4 * kq, w[ob]);
        // softmax backward over the 16 rows, Linear(CS, CS) backward, ReLU / BN2 bookkeeping
        f32x4 gz[NOB];
        if (CS < 16) gw[0] += xchg32(gw[0]);   // 8 units: lanes kq and kq ^ 2 hold partial sums of the same units
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const bool hv = 16 * ob + 4 * kq < CS;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dot = sum16(w[ob][r] * gw[ob][r]);
                gz[ob][r] = hv ? w[ob][r] * (gw[ob][r] - dot) : 0.f;
            }
            sgz[ob] += gz[ob];
            if (hv) {
                st4(gz_t + row * GS + 16 * ob + 4 * kq, gz[ob]);
                st4(u_t + row * GS + 16 * ob + 4 * kq, u[ob]);
            }
        }
#pragma unroll
        for (int ub = 0; ub < NOB; ++ub) {
            f32x4 gu = zero4();   // g_u^T = Ww2^T g_z^T: A operand Ww2[16 jo + 4 kq + e][16 ub + (l & 15)]
#pragma unroll
            for (int jo = 0; jo < NOB; ++jo)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    gu = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[(16 * jo + 4 * kq + e) * WS2 + 16 * ub + row], gz[jo][e], gu, 0, 0, 0);
            f32x4 gy2;
#pragma unroll
            for (int r = 0; r < 4; ++r) gy2[r] = u[ub][r] > 0.f ? gu[r] : 0.f;
            const bool hv = 16 * ub + 4 * kq < CS;
            if (hv) st_row4<BF>(A.G2, ri * CS + 16 * ub + 4 * kq, gy2);
            sg[ub] += gy2;
            sgh[ub] += sel4(hv, gy2 * ((h[ub] - unit4<C>(ucst, U_MEAN, ub, kq)) * unit4<C>(ucst, U_RSTD, ub, kq)));
        }
        wave_sync();
        // g_Ww2[o][u'] += sum_rows g_z[row][o] u[row][u']  (reduction index = rows: operands re-read lanes-along-units)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float a_[NOB], b_[NOB];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) { a_[ob] = gz_t[(4 * t + kq) * GS + 16 * ob + row]; b_[ob] = u_t[(4 * t + kq) * GS + 16 * ob + row]; }
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int ub = 0; ub < NOB; ++ub) accw[ob][ub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[ob], b_[ub], accw[ob][ub], 0, 0, 0);
        }
        wave_sync();
    }
    block_row(lds, W, [&](RowAcc o) {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = sg[ob][r], b = sgh[ob][r], d = sgz[ob][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); d += __shfl_xor(d, m, 64); }
                if (row == 0 && 16 * ob + 4 * kq < CS) { o[16 * ob + 4 * kq + r] = a; o[CS + 16 * ob + 4 * kq + r] = b; o[2 * CS + 16 * ob + 4 * kq + r] = d; }
            }
#pragma unroll
            for (int ub = 0; ub < NOB; ++ub)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * ob + 4 * kq < CS && 16 * ub + row < CS) o[3 * CS + (16 * ob + 4 * kq + r) * CS + 16 * ub + row] = accw[ob][ub][r];
        }
    });
    store_row(lds, W, A.partial + (size_t)blockIdx.x * W);
}

// ------------------------------------------------------------------------------------------------ B2 (64-channel slabs)
// partial row per block (all slabs of one blockIdx.x write disjoint columns of the same row):
//   [sum g_y1 (C) | sum g_y1*rhat (C) | g_bw1 (CS) | g_Ww1 (CS*C)]   (as fl::k_b2)
template <int C, bool BF>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(C == 64 ? 3 : 1, 8))) void k_b2(LayerArgs A) {   // (C = 64: 180 -> 168 registers, 9 spilled: a third wave per SIMD, 80 -> 77 us with 768 workgroups; the same hint costs C = 128 +25 %)
    constexpr int CS = C / 8, NOB = nob_of(C), CSP = csp_of(C), GS = CSP + 4, WS = 68, W = 2 * C + CS + CS * C;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wv = threadIdx.x >> 6;
    float *cst = lds, *ucst = cst + 6 * C, *wl = ucst + 7 * CSP;           // wl: Ww1[:, slab], row stride 68
    float *gh_t = wl + CSP * WS + wv * 16 * GS, *v1_t = wl + CSP * WS + WPB * 16 * GS + wv * 16 * TS;
    const int slab = blockIdx.y, c0 = 64 * slab;
    stage_rows<CSP, 64, WS>(wl, gp(A.Ww1) + c0, C, CS);
    for (int e = threadIdx.x; e < WPB * 16 * GS; e += NT) (wl + CSP * WS)[e] = 0.f;   // g_h tiles incl. padding columns
    stage_consts<C>(cst, A, true);
    stage_units<C, true>(ucst, A, gp(A.sums));
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const GeoW G = geo_weights(A);
    f32x4 sg[4], sgr[4], sgh[NOB], accw[NOB][4], m1[4], r1[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        sg[jj] = zero4(); sgr[jj] = zero4();
        m1[jj] = ld4(gp(A.mean) + 3 + c0 + 16 * jj + 4 * kq);
        r1[jj] = ld4(gp(A.rstd) + 3 + c0 + 16 * jj + 4 * kq);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) accw[ob][jj] = zero4();
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) sgh[ob] = zero4();
    fl::PointWalk pw(A, wv);
    int nb_next = pw.valid() ? A.idx[pw.point() * 16 + row] : -1;
    for (; pw.valid(); pw.step()) {
        const long i = pw.point();
        const size_t ri = (size_t)i * 16 + row;
        const int nb = nb_next;
        const size_t nbc = (size_t)max(nb, 0);
        nb_next = A.idx[(pw.has_next() ? pw.next_point() : i) * 16 + row];   // next trip's index: in flight during this trip
        float pn[3], pi[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) { pn[b] = A.p[nbc * 3 + b]; pi[b] = A.p[(size_t)i * 3 + b]; }
        f32x4 h[NOB], g2[NOB], xk[4], xq[4];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) { h[ob] = ld_row4<BF>(A.H, ri * CS + unit_off<C>(ob, kq)); g2[ob] = ld_row4<BF>(A.G2, ri * CS + unit_off<C>(ob, kq)); }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int c4 = c0 + 16 * jj + 4 * kq;
            xk[jj] = ld4(A.xk + nbc * C + c4); xq[jj] = ld4(A.xq + (size_t)i * C + c4);
        }
        __builtin_amdgcn_sched_barrier(0);
        const Geo R = geo_of(G, nb, pn, pi);
        f32x4 gh[NOB];
        hidden_grad<C>(ucst, kq, h, g2, gh);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            sgh[ob] += gh[ob];
            if (16 * ob + 4 * kq < CS) st4(gh_t + row * GS + 16 * ob + 4 * kq, gh[ob]);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int g = 4 * (4 * slab + jj) + kq;
            f32x4 acc = zero4();   // (Ww1^T g_h)[channel 16 j + 4 kq + reg][row]
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[(16 * ob + 4 * kq + e) * WS + 16 * jj + row], gh[ob][e], acc, 0, 0, 0);
            const f32x4 r = (sel4(nb >= 0, xk[jj]) - xq[jj]) + pos4(cst, C, g, R.t1n);
            const f32x4 y1 = r * ld4(cst + 4 * C + 4 * g) + ld4(cst + 5 * C + 4 * g);
            f32x4 gy1;
#pragma unroll
            for (int e = 0; e < 4; ++e) gy1[e] = y1[e] > 0.f ? acc[e] : 0.f;
            sg[jj] += gy1;
            sgr[jj] += gy1 * ((r - m1[jj]) * r1[jj]);
            st4(v1_t + row * TS + 16 * jj + 4 * kq, relu4(y1));
        }
        wave_sync();
        // g_Ww1[o][c] += sum_rows g_h[row][o] v1[row][c]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float a_[NOB], b_[4];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) a_[ob] = gh_t[(4 * t + kq) * GS + 16 * ob + row];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) b_[jj] = v1_t[(4 * t + kq) * TS + 16 * jj + row];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) accw[ob][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[ob], b_[jj], accw[ob][jj], 0, 0, 0);
        }
        wave_sync();
    }
    // slab-local row in LDS: [sum g_y1 (64) | sum g_y1*rhat (64) | g_bw1 (CSP; slab 0 only) | g_Ww1[:, slab] (CS x 64)]
    block_row(lds, 128 + CSP + CS * 64, [&](RowAcc o) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = sg[jj][r], b = sgr[jj][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
                if (row == 0) { o[16 * jj + 4 * kq + r] = a; o[64 + 16 * jj + 4 * kq + r] = b; }
            }
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            if (slab == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = sgh[ob][r];
#pragma unroll
                    for (int m = 1; m < 16; m <<= 1) a += __shfl_xor(a, m, 64);
                    if (row == 0 && 16 * ob + 4 * kq < CS) o[128 + 16 * ob + 4 * kq + r] = a;
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * ob + 4 * kq < CS) o[128 + CSP + (16 * ob + 4 * kq + r) * 64 + 16 * jj + row] = accw[ob][jj][r];
        }
    });
    float *dst = A.partial + (size_t)blockIdx.x * W;
    if (threadIdx.x < 128) dst[(threadIdx.x >> 6) * C + c0 + (threadIdx.x & 63)] = lds[threadIdx.x];
    if (slab == 0 && threadIdx.x < CS) dst[2 * C + threadIdx.x] = lds[128 + threadIdx.x];
    for (int e = threadIdx.x; e < CS * 64; e += NT) dst[2 * C + CS + (size_t)(e >> 6) * C + c0 + (e & 63)] = lds[128 + CSP + e];
}

// ------------------------------------------------------------------------------------------------ B2 at level 1 (C = 32, nsample 8)
// The same pass for the first level, where the row-per-lane form (fl::k_b2<32,8>) is bound by its vector issue and restores ~500 spilled
// scalar registers per tile: one wave = TWO points = 2 x 8 neighbour rows (rows 0-7 / 8-15 of the 16-row tile; row r of pair pp is flat
// row 16 pp + r of every (N, 8, .) array, i.e. the row-major layout as it is), ONE 32-channel slab (two 16-channel blocks), the 4 hidden
// units in the kq = 0 lanes of a 16-unit block (12 rows of zero padding in the A operand).  Everything per row is as in k_b2; only the
// centre point of a lane's row (x_q, coordinates) is 2 pp + (row >> 3), and an odd N leaves the upper half of the last tile empty.
// partial row: [sum g_y1 (32) | sum g_y1*rhat (32) | g_bw1 (4) | g_Ww1 (4 x 32)]   (= fl::k_b2<32, 8>'s row)
template <bool BF>
__global__ __launch_bounds__(64 * WPB) void k_b2_l1(LayerArgs A) {
    constexpr int C = 32, CS = 4, CSP = 16, GS = CSP + 4, CW = 32, NJ = 2, WS = CW + 4, W = 2 * C + CS + CS * C;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wv = threadIdx.x >> 6;
    float *cst = lds, *ucst = cst + 6 * C, *wl = ucst + 7 * CSP;           // wl: Ww1 (zero rows 4 .. 15), row stride 36
    float *gh_t = wl + CSP * WS + wv * 16 * GS, *v1_t = wl + CSP * WS + WPB * 16 * GS + wv * 16 * TS;
    stage_rows<CSP, CW, WS>(wl, gp(A.Ww1), C, CS);
    for (int e = threadIdx.x; e < WPB * 16 * GS; e += NT) (wl + CSP * WS)[e] = 0.f;   // g_h tiles incl. padding columns
    stage_consts<C>(cst, A, true);
    stage_units<C, true>(ucst, A, gp(A.sums));
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const GeoW G = geo_weights(A);
    f32x4 sg[NJ], sgr[NJ], sgh = zero4(), accw[NJ], m1[NJ], r1[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        sg[jj] = zero4(); sgr[jj] = zero4(); accw[jj] = zero4();
        m1[jj] = ld4(gp(A.mean) + 3 + 16 * jj + 4 * kq);
        r1[jj] = ld4(gp(A.rstd) + 3 + 16 * jj + 4 * kq);
    }
    LayerArgs Ap = A;
    Ap.N = (A.N + 1) / 2;          // the walk is over PAIRS of points
    Ap.order = nullptr;
    fl::PointWalk pw(Ap, wv);
    const long last_row = (long)A.N * 8 - 1;
    auto row_of = [&](long pp) { return min(pp * 16 + row, last_row); };   // (clamped: the upper half of an odd N's last tile)
    int nb_next = pw.valid() ? A.idx[row_of(pw.point())] : -1;
    for (; pw.valid(); pw.step()) {
        const long pp = pw.point(), i = 2 * pp + (row >> 3);
        const bool valid = i < A.N;
        const size_t ri = (size_t)row_of(pp), ic = (size_t)min(i, (long)A.N - 1);
        const int nb = valid ? nb_next : -1;
        const size_t nbc = (size_t)max(nb, 0);
        nb_next = A.idx[row_of(pw.has_next() ? pw.next_point() : pp)];   // next trip's index: in flight during this trip
        float pn[3], pi[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) { pn[b] = A.p[nbc * 3 + b]; pi[b] = A.p[ic * 3 + b]; }
        f32x4 h[1], g2[1], xk[NJ], xq[NJ];
        h[0] = ld_row4<BF>(A.H, ri * CS); g2[0] = ld_row4<BF>(A.G2, ri * CS);
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int c4 = 16 * jj + 4 * kq;
            xk[jj] = ld4(A.xk + nbc * C + c4); xq[jj] = ld4(A.xq + ic * C + c4);
        }
        __builtin_amdgcn_sched_barrier(0);
        const Geo R = geo_of(G, nb, pn, pi);
        f32x4 gh[1];
        hidden_grad<C>(ucst, kq, h, g2, gh);
        gh[0] = sel4(valid, gh[0]);
        sgh += gh[0];
        if (kq == 0) st4(gh_t + row * GS, gh[0]);
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int g = 4 * jj + kq;
            f32x4 acc = zero4();   // (Ww1^T g_h)[channel 16 j + 4 kq + reg][row]
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[(4 * kq + e) * WS + 16 * jj + row], gh[0][e], acc, 0, 0, 0);
            const f32x4 r = (sel4(nb >= 0, xk[jj]) - xq[jj]) + pos4(cst, C, g, R.t1n);
            const f32x4 y1 = r * ld4(cst + 4 * C + 4 * g) + ld4(cst + 5 * C + 4 * g);
            f32x4 gy1;
#pragma unroll
            for (int e = 0; e < 4; ++e) gy1[e] = y1[e] > 0.f ? acc[e] : 0.f;
            sg[jj] += gy1;
            sgr[jj] += gy1 * ((r - m1[jj]) * r1[jj]);
            st4(v1_t + row * TS + 16 * jj + 4 * kq, relu4(y1));
        }
        wave_sync();
        // g_Ww1[o][c] += sum_rows g_h[row][o] v1[row][c]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float a_ = gh_t[(4 * t + kq) * GS + row];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) accw[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, v1_t[(4 * t + kq) * TS + 16 * jj + row], accw[jj], 0, 0, 0);
        }
        wave_sync();
    }
    // row in LDS: [sum g_y1 (32) | sum g_y1*rhat (32) | g_bw1 (16: 4 used) | g_Ww1 (4 x 32)]
    block_row(lds, 2 * CW + CSP + CS * CW, [&](RowAcc o) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = sg[jj][r], b = sgr[jj][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
                if (row == 0) { o[16 * jj + 4 * kq + r] = a; o[CW + 16 * jj + 4 * kq + r] = b; }
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = sgh[r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) a += __shfl_xor(a, m, 64);
            if (row == 0 && kq == 0) o[2 * CW + r] = a;
        }
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kq == 0) o[2 * CW + CSP + r * CW + 16 * jj + row] = accw[jj][r];
    });
    float *dst = A.partial + (size_t)blockIdx.x * W;
    if (threadIdx.x < 2 * C) dst[threadIdx.x] = lds[threadIdx.x];
    if (threadIdx.x < CS) dst[2 * C + threadIdx.x] = lds[2 * CW + threadIdx.x];
    if (threadIdx.x < CS * C) dst[2 * C + CS + threadIdx.x] = lds[2 * CW + CSP + threadIdx.x];
}

// ------------------------------------------------------------------------------------------------ B3
// partial row per block: [sum g_yp (3) | sum g_yp*that (3) | pad 2 | g_bp2 (C) | g_Wp2 (C*3) | sum g_yp (x) rel (9) | pad 7]   (as fl::k_b3)
//
// Round 4: the POINTS are the outer loop and the 64-channel chunks the inner one (rounds 1-3: chunks outermost, every chunk re-staged its
// Ww1 slab behind two block barriers and re-read the point's H / G2 / Wsm / coordinates / G3 -- at levels 4-5, where a wave sees one or
// two points, the launch was a chain of NCHK x (stage, barrier, trip, barrier): 47 us for 3,124 points).  Up to four chunks (256 channels)
// of Ww1 and of the per-channel constants are staged at once; a point's row data is loaded once, its chunks run back to back (the next
// chunk's rows are requested before the current chunk's column phase), g_t1n stays in registers and G3 is written once.  C = 512 takes two
// such sweeps over the points (the second adds to the first one's G3), the other widths one.
template <int C> constexpr int b3_ncp() { return C / 64 < 4 ? C / 64 : 4; }          // chunks per sweep
template <int C> constexpr int b3_wls() { return 64 * b3_ncp<C>() + 4; }             // row stride of the Ww1 copy (floats)
template <int C> constexpr bool b3_gacc_lds() { return C == 128 || C == 512; }       // geometry-branch sums in LDS slots (else registers)
template <int C, bool BF>
__global__ __launch_bounds__(64 * WPB) void k_b3(LayerArgs A) {
    constexpr int CS = C / 8, NOB = nob_of(C), CSP = csp_of(C), NCHK = C / 64, NCP = b3_ncp<C>(), NSW = NCHK / NCP, WS = b3_wls<C>(), W = 8 + 4 * C + 16;
    constexpr int CW = 64 * NCP;             // channels of a sweep
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wv = threadIdx.x >> 6;
    float *cst = lds;                        // Wp2 (3C, channel-major) | bp2 (C) | s1 (C) | t1 (C)
    float *ccst = cst + 6 * C;               // the sweep's channels: mean1 | rstd1 | sum g_y1 / rows | sum g_y1*rhat / rows   (4 x CW)
    float *ucst = ccst + 4 * CW;             // per-unit constants (stage_units), sums = B1's
    float *wl = ucst + 7 * CSP;              // Ww1[:, sweep's channels], row stride WS
    float *tiles = wl + CSP * WS;
    float *tile = tiles + wv * 32 * TS, *tile2 = tile + 16 * TS;   // g_r tile, g_pr tile (one 64-channel chunk)
    float *t1nt = tiles + WPB * 32 * TS + wv * 64;
    float *crow = tiles;                     // epilogue: WPB x 4 CW columns (the tiles are free then)
    stage_consts<C>(cst, A, true);
    const float *S2 = gp(A.sums);   // [sum g_y1 (C) | sum g_y1*rhat (C)]   (A.sums2 = B1's [sum g_y2 | sum g_y2*hhat])
    stage_units<C, true>(ucst, A, gp(A.sums2));
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const GeoW G = geo_weights(A);
    // the 15 sums of the geometry branch [sum g_yp (3) | sum g_yp*that (3) | sum g_yp[a] * rel[b] (9): closed-form BNp backward, fl::k_colsum].
    // C = 128 / 512 sit at the 256-register step of 2 waves per SIMD: one private LDS slot per row-owning lane (stride 15: conflict-free).
    // C = 64 / 256 have the registers, and C = 256 does NOT have the LDS: the 3.8 KB of slots took its block from 80.3 to 84.1 KB, one block
    // per CU instead of two (60 us instead of 41 per launch, found in the round-4 kernel trace) -- there the sums stay in registers.
    constexpr bool GLDS = b3_gacc_lds<C>();
    float *gacc = tiles + WPB * 32 * TS + WPB * 64 + (wv * 16 + row) * 15;
    float greg[15];
#pragma unroll
    for (int e = 0; e < 15; ++e) greg[e] = 0.f;
    if (GLDS && kq == 0) {
#pragma unroll
        for (int e = 0; e < 15; ++e) gacc[e] = 0.f;
    }
    float *dst = A.partial + (size_t)blockIdx.x * W;   // the block's partial row
#pragma unroll 1
    for (int sw = 0; sw < NSW; ++sw) {
        const int cb = CW * sw;              // first channel of the sweep
        __syncthreads();                     // previous sweep fully consumed (and, first sweep, constants staged)
        stage_rows<CSP, CW, WS>(wl, gp(A.Ww1) + cb, C, CS);
        for (int e = threadIdx.x; e < 4 * CW; e += NT) {
            const int arr = e / CW, c = cb + e % CW;
            const float *src = arr == 0 ? gp(A.mean) + 3 : arr == 1 ? gp(A.rstd) + 3 : arr == 2 ? S2 : S2 + C;
            ccst[e] = src[c] * (arr >= 2 ? A.inv_rows : 1.f);
        }
        __syncthreads();
        float sbp2[NCP], awp2[NCP][3];
#pragma unroll
        for (int qc = 0; qc < NCP; ++qc) { sbp2[qc] = 0.f; awp2[qc][0] = 0.f; awp2[qc][1] = 0.f; awp2[qc][2] = 0.f; }
        fl::PointWalk pw(A, wv);
        int nb_next = pw.valid() ? A.idx[pw.point() * 16 + row] : -1;
        for (; pw.valid(); pw.step()) {
            const long i = pw.point();
            const size_t ri = (size_t)i * 16 + row;
            const int nb = nb_next;
            const size_t nbc = (size_t)max(nb, 0);
            // ---------------- every global load of the point's row data + its first chunk
            nb_next = A.idx[(pw.has_next() ? pw.next_point() : i) * 16 + row];
            float pn[3], pi[3], g3old[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) { pn[b] = A.p[nbc * 3 + b]; pi[b] = A.p[(size_t)i * 3 + b]; g3old[b] = NSW > 1 ? A.G3[ri * 3 + b] : 0.f; }
            f32x4 hh[NOB], g2[NOB], w[NOB], xk[4], xq[4], go[4];
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
                const size_t o = ri * CS + unit_off<C>(ob, kq);
                hh[ob] = ld_row4<BF>(A.H, o); g2[ob] = ld_row4<BF>(A.G2, o); w[ob] = ld_row4<BF>(A.Wsm, o);
            }
            const float *xkr = A.xk + nbc * C + cb + 4 * kq, *xqr = A.xq + (size_t)i * C + cb + 4 * kq, *gor = A.gout + (size_t)i * C + cb + 4 * kq;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { xk[jj] = ld4(xkr + 16 * jj); xq[jj] = ld4(xqr + 16 * jj); go[jj] = ld4(gor + 16 * jj); }
            __builtin_amdgcn_sched_barrier(0);
            const Geo R = geo_of(G, nb, pn, pi);
            if (kq == 0) { t1nt[row * 4 + 0] = R.t1n[0]; t1nt[row * 4 + 1] = R.t1n[1]; t1nt[row * 4 + 2] = R.t1n[2]; }
            // g_h of the lane's hidden units (BN2 backward); softmax weights as B1 stored them (Wsm)
            f32x4 gh[NOB];
            hidden_grad<C>(ucst, kq, hh, g2, gh);
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) w[ob] = sel4(16 * ob + 4 * kq < CS, w[ob]);
            if (CS < 16) {   // channels of lanes kq = 2, 3 use the units of lanes kq - 2
                const f32x4 x = xchg32(w[0]);
                if (kq >= 2) w[0] = x;
            }
            float gt1n[3] = {0.f, 0.f, 0.f};
            // (rolled on purpose: unrolled, the compiler keeps the operands of all four chunks live -- 512 registers and spills at C >= 128)
#pragma unroll 1
            for (int qc = 0; qc < NCP; ++qc) {
                const int q = NCP * sw + qc;                                  // chunk in the layer
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int g = 4 * (4 * q + jj) + kq, gl = 64 * qc + 4 * (4 * jj + kq);   // channel group in the layer / first channel in the sweep
                    f32x4 acc = zero4();
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[(16 * ob + 4 * kq + e) * WS + 64 * qc + 16 * jj + row], gh[ob][e], acc, 0, 0, 0);
                    const f32x4 r = (sel4(nb >= 0, xk[jj]) - xq[jj]) + pos4(cst, C, g, R.t1n);
                    const f32x4 s1 = ld4(cst + 4 * C + 4 * g);
                    const f32x4 y1 = r * s1 + ld4(cst + 5 * C + 4 * g);
                    f32x4 gy1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gy1[e] = y1[e] > 0.f ? acc[e] : 0.f;
                    // BN1 backward: g_r = s1 * (g_y1 - mean(g_y1) - rhat * mean(g_y1 * rhat))
                    const f32x4 rhat = (r - ld4(ccst + gl)) * ld4(ccst + CW + gl);
                    const f32x4 gr = s1 * (gy1 - ld4(ccst + 2 * CW + gl) - rhat * ld4(ccst + 3 * CW + gl));
                    st4(tile + row * TS + 16 * jj + 4 * kq, gr);
                    const f32x4 gpr = gr + go[jj] * w[jj % NOB];   // + the aggregation's share of p_r  ((4 q + jj) mod NOB = jj mod NOB: NOB divides 4)
                    st4(tile2 + row * TS + 16 * jj + 4 * kq, gpr);
                    const f32x4 w0 = ld4(cst + 12 * g), w1 = ld4(cst + 12 * g + 4), w2v = ld4(cst + 12 * g + 8);
                    const float wp[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2v[0], w2v[1], w2v[2], w2v[3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        gt1n[0] += gpr[e] * wp[3 * e]; gt1n[1] += gpr[e] * wp[3 * e + 1]; gt1n[2] += gpr[e] * wp[3 * e + 2];
                    }
                }
                {   // the next chunk's rows: requested now, in flight during this chunk's column phase (last chunk: re-reads itself, unused)
                    const int qn = qc + 1 < NCP ? qc + 1 : qc;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        xk[jj] = ld4(xkr + 64 * qn + 16 * jj); xq[jj] = ld4(xqr + 64 * qn + 16 * jj); go[jj] = ld4(gor + 64 * qn + 16 * jj);
                    }
                }
                wave_sync();
                {   // lanes along channels: g_r rows out (256 B per row and chunk), g_xq[i] = - sum_rows g_r
                    float acc = 0.f;
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) {
                        const float v = tile[rr * TS + lane];
                        acc += v;
                        const size_t o = ((size_t)i * 16 + rr) * C + 64 * q + lane;   // g_xk = segmented sum of these rows
                        if constexpr (BF) __builtin_nontemporal_store((unsigned short)fl::f2bf(v), reinterpret_cast<unsigned short *>(A.GR) + o);
                        else __builtin_nontemporal_store(v, A.GR + o);
                    }
                    A.gxq[(size_t)i * C + 64 * q + lane] = -acc;
                }
                float cb2 = 0.f, cw2[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {   // g_bp2 / g_Wp2 of channel 64 q + lane
                    const float v = tile2[rr * TS + lane];
                    cb2 += v;
                    cw2[0] += v * t1nt[rr * 4 + 0]; cw2[1] += v * t1nt[rr * 4 + 1]; cw2[2] += v * t1nt[rr * 4 + 2];
                }
#pragma unroll
                for (int k = 0; k < NCP; ++k) {   // (static register indices: the chunk's sums go to accumulator set qc through selects)
                    const bool on = k == qc;
                    sbp2[k] += on ? cb2 : 0.f;
                    awp2[k][0] += on ? cw2[0] : 0.f; awp2[k][1] += on ? cw2[1] : 0.f; awp2[k][2] += on ? cw2[2] : 0.f;
                }
                wave_sync();
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float v = gt1n[a];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (kq == 0) {
                    if (sw > 0) v += g3old[a];
                    if (sw == NSW - 1) {   // all channels seen: ReLU mask of BNp, BNp-backward sums
                        v = R.t1n[a] > 0.f ? v : 0.f;
                        const float vt = v * ((R.t1[a] - A.mean[a]) * A.rstd[a]);
                        if constexpr (GLDS) {
                            gacc[a] += v;
                            gacc[3 + a] += vt;
                            gacc[6 + 3 * a + 0] += v * R.rel[0]; gacc[6 + 3 * a + 1] += v * R.rel[1]; gacc[6 + 3 * a + 2] += v * R.rel[2];
                        } else {
                            greg[a] += v;
                            greg[3 + a] += vt;
                            greg[6 + 3 * a + 0] += v * R.rel[0]; greg[6 + 3 * a + 1] += v * R.rel[1]; greg[6 + 3 * a + 2] += v * R.rel[2];
                        }
                    }
                    A.G3[ri * 3 + a] = v;
                }
            }
        }
        // the sweep's columns of the block's row (g_bp2: CW, g_Wp2: CW x 3): per wave into LDS (the tiles are free), summed in wave order
        __syncthreads();
#pragma unroll
        for (int qc = 0; qc < NCP; ++qc) {
            crow[wv * 4 * CW + 64 * qc + lane] = sbp2[qc];
#pragma unroll
            for (int a = 0; a < 3; ++a) crow[wv * 4 * CW + CW + (64 * qc + lane) * 3 + a] = awp2[qc][a];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 4 * CW; t += NT) {
            float v = crow[t];
#pragma unroll
            for (int w = 1; w < WPB; ++w) v += crow[w * 4 * CW + t];
            if (t < CW) dst[8 + cb + t] = v; else dst[8 + C + (size_t)3 * cb + (t - CW)] = v;
        }
    }
    __syncthreads();
    block_row(crow, 24, [&](RowAcc o) {
#pragma unroll
        for (int e = 0; e < 15; ++e) {
            const float x = pdf_wave_sum_f32(kq == 0 ? (GLDS ? gacc[e] : greg[e]) : 0.f);
            if (lane == 0) o[e < 6 ? e : e + 2] = x;
        }
        if (lane == 0) { o[6] = 0.f; o[7] = 0.f; for (int e = 17; e < 24; ++e) o[e] = 0.f; }
    });
    store_row(crow, 8, dst);
    if (threadIdx.x < 16) dst[8 + 4 * C + threadIdx.x] = crow[8 + threadIdx.x];
}

// ------------------------------------------------------------------------------------------------ B3 at level 1 (C = 32, nsample 8)
// Two points per wave as in k_b2_l1, ONE 32-channel chunk (no sweeps, no chunk loop).  The four hidden units of a row are the four values
// every lane loads (a lane's channels 16 j + 4 kq + e use unit e): no exchange between the kq groups.  The column phase runs lanes along
// (point half, channel): lane l sums the 8 rows of point 2 pp + (l >> 5) for channel l & 31 -- the g_r rows leave as two 128-byte rows
// per step, g_xq of both points in one store, g_bp2 / g_Wp2 as per-half partial sums that meet in the epilogue.
// partial row: [sum g_yp (3) | sum g_yp*that (3) | pad 2 | g_bp2 (32) | g_Wp2 (32 x 3) | sum g_yp (x) rel (9) | pad 7]   (= fl::k_b3<32, 8>'s row)
template <bool BF>
__global__ __launch_bounds__(64 * WPB) void k_b3_l1(LayerArgs A) {
    constexpr int C = 32, CS = 4, CSP = 16, CW = 32, NJ = 2, WS = CW + 4, W = 8 + 4 * C + 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wv = threadIdx.x >> 6;
    float *cst = lds;                        // Wp2 (3C, channel-major) | bp2 (C) | s1 (C) | t1 (C)
    float *ccst = cst + 6 * C;               // mean1 | rstd1 | sum g_y1 / rows | sum g_y1*rhat / rows   (4 x 32)
    float *ucst = ccst + 4 * CW;             // per-unit constants (stage_units), sums = B1's
    float *wl = ucst + 7 * CSP;              // Ww1 (zero rows 4 .. 15), row stride 36
    float *tiles = wl + CSP * WS;
    float *tile = tiles + wv * 32 * TS, *tile2 = tile + 16 * TS;   // g_r tile, g_pr tile (16 rows x 32 channels)
    float *t1nt = tiles + WPB * 32 * TS + wv * 64;
    float *crow = tiles;                     // epilogue: WPB x 2 x 4 CW columns (the tiles are free then)
    stage_consts<C>(cst, A, true);
    const float *S2 = gp(A.sums);            // [sum g_y1 (C) | sum g_y1*rhat (C)]   (A.sums2 = B1's)
    stage_units<C, true>(ucst, A, gp(A.sums2));
    stage_rows<CSP, CW, WS>(wl, gp(A.Ww1), C, CS);
    for (int e = threadIdx.x; e < 4 * CW; e += NT) {
        const int arr = e / CW, c = e % CW;
        const float *src = arr == 0 ? gp(A.mean) + 3 : arr == 1 ? gp(A.rstd) + 3 : arr == 2 ? S2 : S2 + C;
        ccst[e] = src[c] * (arr >= 2 ? A.inv_rows : 1.f);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4, half = lane >> 5, ch = lane & 31;
    const GeoW G = geo_weights(A);
    float greg[15];
#pragma unroll
    for (int e = 0; e < 15; ++e) greg[e] = 0.f;
    float sbp2 = 0.f, awp2[3] = {0.f, 0.f, 0.f};
    LayerArgs Ap = A;
    Ap.N = (A.N + 1) / 2;          // the walk is over PAIRS of points
    Ap.order = nullptr;
    fl::PointWalk pw(Ap, wv);
    const long last_row = (long)A.N * 8 - 1;
    auto row_of = [&](long pp) { return min(pp * 16 + row, last_row); };
    int nb_next = pw.valid() ? A.idx[row_of(pw.point())] : -1;
    for (; pw.valid(); pw.step()) {
        const long pp = pw.point(), i = 2 * pp + (row >> 3);
        const bool valid = i < A.N;
        const size_t ri = (size_t)row_of(pp), ic = (size_t)min(i, (long)A.N - 1);
        const int nb = valid ? nb_next : -1;
        const size_t nbc = (size_t)max(nb, 0);
        nb_next = A.idx[row_of(pw.has_next() ? pw.next_point() : pp)];
        float pn[3], pi[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) { pn[b] = A.p[nbc * 3 + b]; pi[b] = A.p[ic * 3 + b]; }
        f32x4 hh[1], g2[1], xk[NJ], xq[NJ], go[NJ];
        hh[0] = ld_row4<BF>(A.H, ri * CS); g2[0] = ld_row4<BF>(A.G2, ri * CS);
        const f32x4 w = ld_row4<BF>(A.Wsm, ri * CS);     // softmax weights of the row's 4 units as B1 stored them
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int c4 = 16 * jj + 4 * kq;
            xk[jj] = ld4(A.xk + nbc * C + c4); xq[jj] = ld4(A.xq + ic * C + c4); go[jj] = ld4(A.gout + ic * C + c4);
        }
        __builtin_amdgcn_sched_barrier(0);
        const Geo R = geo_of(G, nb, pn, pi);
        if (kq == 0) { t1nt[row * 4 + 0] = valid ? R.t1n[0] : 0.f; t1nt[row * 4 + 1] = valid ? R.t1n[1] : 0.f; t1nt[row * 4 + 2] = valid ? R.t1n[2] : 0.f; }
        f32x4 gh[1];
        hidden_grad<C>(ucst, kq, hh, g2, gh);
        float gt1n[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int g = 4 * jj + kq, gl = 4 * g;   // channel group / first channel
            f32x4 acc = zero4();
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[(4 * kq + e) * WS + 16 * jj + row], gh[0][e], acc, 0, 0, 0);
            const f32x4 r = (sel4(nb >= 0, xk[jj]) - xq[jj]) + pos4(cst, C, g, R.t1n);
            const f32x4 s1 = ld4(cst + 4 * C + 4 * g);
            const f32x4 y1 = r * s1 + ld4(cst + 5 * C + 4 * g);
            f32x4 gy1;
#pragma unroll
            for (int e = 0; e < 4; ++e) gy1[e] = y1[e] > 0.f ? acc[e] : 0.f;
            // BN1 backward: g_r = s1 * (g_y1 - mean(g_y1) - rhat * mean(g_y1 * rhat))
            const f32x4 rhat = (r - ld4(ccst + gl)) * ld4(ccst + CW + gl);
            const f32x4 gr = sel4(valid, s1 * (gy1 - ld4(ccst + 2 * CW + gl) - rhat * ld4(ccst + 3 * CW + gl)));
            st4(tile + row * TS + 16 * jj + 4 * kq, gr);
            const f32x4 gpr = gr + sel4(valid, go[jj] * w);   // + the aggregation's share of p_r (channel 4 g + e uses unit e)
            st4(tile2 + row * TS + 16 * jj + 4 * kq, gpr);
            const f32x4 w0 = ld4(cst + 12 * g), w1 = ld4(cst + 12 * g + 4), w2v = ld4(cst + 12 * g + 8);
            const float wp[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2v[0], w2v[1], w2v[2], w2v[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gt1n[0] += gpr[e] * wp[3 * e]; gt1n[1] += gpr[e] * wp[3 * e + 1]; gt1n[2] += gpr[e] * wp[3 * e + 2];
            }
        }
        wave_sync();
        {   // lanes along (point half, channel): g_r rows out, g_xq[i] = - sum of the point's 8 rows
            const long ip = 2 * pp + half;
            const bool pv = ip < A.N;
            float acc = 0.f;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const float v = tile[(8 * half + rr) * TS + ch];
                acc += v;
                if (pv) {
                    const size_t o = ((size_t)ip * 8 + rr) * C + ch;   // g_xk = segmented sum of these rows
                    if constexpr (BF) __builtin_nontemporal_store((unsigned short)fl::f2bf(v), reinterpret_cast<unsigned short *>(A.GR) + o);
                    else __builtin_nontemporal_store(v, A.GR + o);
                }
            }
            if (pv) A.gxq[(size_t)ip * C + ch] = -acc;
            float cb2 = 0.f, cw2[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {   // g_bp2 / g_Wp2 of channel ch: this half's 8 rows
                const float v = tile2[(8 * half + rr) * TS + ch];
                cb2 += v;
                cw2[0] += v * t1nt[(8 * half + rr) * 4 + 0]; cw2[1] += v * t1nt[(8 * half + rr) * 4 + 1]; cw2[2] += v * t1nt[(8 * half + rr) * 4 + 2];
            }
            sbp2 += cb2; awp2[0] += cw2[0]; awp2[1] += cw2[1]; awp2[2] += cw2[2];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = gt1n[a];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0) {
                v = (valid && R.t1n[a] > 0.f) ? v : 0.f;
                greg[a] += v;
                greg[3 + a] += v * ((R.t1[a] - A.mean[a]) * A.rstd[a]);
                greg[6 + 3 * a + 0] += v * R.rel[0]; greg[6 + 3 * a + 1] += v * R.rel[1]; greg[6 + 3 * a + 2] += v * R.rel[2];
                if (valid) A.G3[ri * 3 + a] = v;
            }
        }
        wave_sync();
    }
    // columns of the block's row (g_bp2: 32, g_Wp2: 32 x 3): per wave and point half into LDS, summed in (wave, half) order
    float *dst = A.partial + (size_t)blockIdx.x * W;
    __syncthreads();
    crow[(2 * wv + half) * 4 * CW + ch] = sbp2;
#pragma unroll
    for (int a = 0; a < 3; ++a) crow[(2 * wv + half) * 4 * CW + CW + ch * 3 + a] = awp2[a];
    __syncthreads();
    for (int t = threadIdx.x; t < 4 * CW; t += NT) {
        float v = crow[t];
#pragma unroll
        for (int w = 1; w < 2 * WPB; ++w) v += crow[w * 4 * CW + t];
        if (t < CW) dst[8 + t] = v; else dst[8 + C + (t - CW)] = v;
    }
    __syncthreads();
    block_row(crow, 24, [&](RowAcc o) {
#pragma unroll
        for (int e = 0; e < 15; ++e) {
            const float x = pdf_wave_sum_f32(kq == 0 ? greg[e] : 0.f);
            if (lane == 0) o[e < 6 ? e : e + 2] = x;
        }
        if (lane == 0) { o[6] = 0.f; o[7] = 0.f; for (int e = 17; e < 24; ++e) o[e] = 0.f; }
    });
    store_row(crow, 8, dst);
    if (threadIdx.x < 16) dst[8 + 4 * C + threadIdx.x] = crow[8 + threadIdx.x];
}

// ------------------------------------------------------------------------------------------------ launchers
// (every allocation also holds the WPB partial rows of the epilogue, fl::block_row)
template <typename KernelT>
static void launch(KernelT kernel, dim3 grid, size_t lds_floats, const LayerArgs &A, hipStream_t s) {
    const size_t lds = lds_floats * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kernel<<<grid, 64 * WPB, lds, s>>>(A);
}
// KERNEL<C, ..., bf16 storage?> by the runtime channel count and storage flag
#define PDF_FLM_C(KERNEL_OF, GRID, LDS)                                             \
    do {                                                                             \
        if (c == 64) { KERNEL_OF(64); } else if (c == 128) { KERNEL_OF(128); }       \
        else if (c == 256) { KERNEL_OF(256); } else { KERNEL_OF(512); }              \
    } while (0)

void launch_p2(const LayerArgs &A, int c, int grid, hipStream_t s) {
    const dim3 g(grid, c / 64);
    const size_t lds = std::max<size_t>((size_t)4 * c, WPB * 128);
#define K_(C_) launch(k_p2<C_>, g, lds, A, s)
    PDF_FLM_C(K_, g, lds);
#undef K_
}
void launch_p3(const LayerArgs &A, int c, bool stats, int grid, hipStream_t s) {
    const dim3 g(grid);
    const size_t lds = (size_t)6 * c + (size_t)csp_of(c) * (c + 4);
#define K_(C_) do { if (stats) { if (A.bf16) launch(k_p3<C_, true, true>, g, lds, A, s); else launch(k_p3<C_, true, false>, g, lds, A, s); } \
                    else { if (A.bf16) launch(k_p3<C_, false, true>, g, lds, A, s); else launch(k_p3<C_, false, false>, g, lds, A, s); } } while (0)
    PDF_FLM_C(K_, g, lds);
#undef K_
}
int p4_grid(long n, int grid) {
    // sized for occupancy (one point per wave and trip); with statistics one partial row of 2 C floats per block
    long gl = (n + WPB - 1) / WPB;
    gl = gl > 2048 ? 2048 : (gl < grid ? grid : gl);
    return (int)gl;
}
void launch_p4(const LayerArgs &A, int c, int grid, bool stats, hipStream_t s) {
    const dim3 g((unsigned)p4_grid(A.N, grid));
    const size_t lds = std::max<size_t>((size_t)4 * c + 3 * csp_of(c) + w2_floats(c), stats ? (size_t)WPB * 2 * c : 0);
#define K_(C_) do { if (stats) { if (A.bf16) launch(k_p4<C_, true, true>, g, lds, A, s); else launch(k_p4<C_, false, true>, g, lds, A, s); } \
                    else { if (A.bf16) launch(k_p4<C_, true, false>, g, lds, A, s); else launch(k_p4<C_, false, false>, g, lds, A, s); } } while (0)
    PDF_FLM_C(K_, g, lds);
#undef K_
}
void launch_b1(const LayerArgs &A, int c, int grid, hipStream_t s) {
    const dim3 g(grid);
    const size_t lds = std::max<size_t>((size_t)4 * c + 7 * csp_of(c) + w2_floats(c) + 2 * WPB * 16 * (csp_of(c) + 4),
                                        (size_t)WPB * (3 * (c / 8) + (c / 8) * (c / 8)));
#define K_(C_) do { if (A.bf16) launch(k_b1<C_, true>, g, lds, A, s); else launch(k_b1<C_, false>, g, lds, A, s); } while (0)
    PDF_FLM_C(K_, g, lds);
#undef K_
}
void launch_b2(const LayerArgs &A, int c, int grid, hipStream_t s) {
    const dim3 g(grid, c / 64);
    const size_t lds = std::max<size_t>((size_t)6 * c + 7 * csp_of(c) + (size_t)csp_of(c) * 68 + WPB * 16 * (csp_of(c) + 4) + WPB * 16 * TS,
                                        (size_t)WPB * (128 + csp_of(c) + c / 8 * 64));
#define K_(C_) do { if (A.bf16) launch(k_b2<C_, true>, g, lds, A, s); else launch(k_b2<C_, false>, g, lds, A, s); } while (0)
    PDF_FLM_C(K_, g, lds);
#undef K_
}
// level 1 (C = 32, nsample 8): PDFOPS_PT_L1_MFMA=0 keeps the row-per-lane passes
bool supported_l1(int nsample, int c) {
    static const bool on = [] { const char *v = getenv("PDFOPS_PT_L1_MFMA"); return !(v && v[0] == '0'); }();
    return on && nsample == 8 && c == 32 && getenv("PDFOPS_PT_NO_MFMA") == nullptr;
}
void launch_b2_l1(const LayerArgs &A, int grid, hipStream_t s) {
    const size_t lds = std::max<size_t>((size_t)6 * 32 + 7 * 16 + (size_t)16 * 36 + WPB * 16 * 20 + WPB * 16 * TS, (size_t)WPB * (2 * 32 + 16 + 4 * 32));
    if (A.bf16) launch(k_b2_l1<true>, dim3(grid), lds, A, s); else launch(k_b2_l1<false>, dim3(grid), lds, A, s);
}
void launch_b3_l1(const LayerArgs &A, int grid, hipStream_t s) {
    const size_t lds = (size_t)6 * 32 + 4 * 32 + 7 * 16 + (size_t)16 * 36 + std::max<size_t>(WPB * 32 * TS + WPB * 64, (size_t)2 * WPB * 4 * 32);
    if (A.bf16) launch(k_b3_l1<true>, dim3(grid), lds, A, s); else launch(k_b3_l1<false>, dim3(grid), lds, A, s);
}
void launch_b3(const LayerArgs &A, int c, int grid, hipStream_t s) {
    const dim3 g(grid);
    const int ncp = c / 64 < 4 ? c / 64 : 4, cw = 64 * ncp;   // (b3_ncp / the sweep's channels)
    const size_t lds = (size_t)6 * c + 4 * cw + 7 * csp_of(c) + (size_t)csp_of(c) * (cw + 4) + std::max<size_t>(WPB * 32 * TS + WPB * 64 + ((c == 128 || c == 512) ? WPB * 240 : 0), (size_t)WPB * 4 * cw);   // (+ the geometry sums' slots, b3_gacc_lds)
#define K_(C_) do { if (A.bf16) launch(k_b3<C_, true>, g, lds, A, s); else launch(k_b3<C_, false>, g, lds, A, s); } while (0)
    PDF_FLM_C(K_, g, lds);
#undef K_
}
#undef PDF_FLM_C

}  // namespace flm
