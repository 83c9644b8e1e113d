// Fused PointTransformerLayer passes, SLAB form (round 6): one wave = one 64-channel slab of one point.
//
// fused_layer_mfma.hip gives a wave ONE point and walks the point's C / 64 chunks one after the other: at levels 4 - 5 (3,124 / 780
// points, C = 256 / 512) that is one or two points per wave, each a serial chain of 4 - 8 chunks behind a block-wide staging of the
// C x C/8 weights -- ~50 us per pass whatever N is (profiles/r05_z_kernel_trace_stats.txt), two waves per SIMD, a third of a wave's
// instructions LDS traffic (weight operands, transposition tiles) and column phases.  Here
//   * a workgroup's waves are the SLABS of one point stream (C >= 256: wave w = slab w; C = 128: two point lanes x two slabs; C = 64:
//     four point lanes): the dependent chain of a point shrinks C/64-fold; 3 waves per SIMD at C <= 256 (<= 168 registers), 2 at 512;
//   * a wave keeps ITS slab of Ww1 in registers for the whole launch (CSP x 64 values = 16 NOB registers: no LDS copy of the weights,
//     no block barrier in the point loop, no sweeps at C = 512) and its 64 channels' constants in a private LDS strip;
//   * row-invariant operands (x_q, g_out of the point) are loaded ONE value per lane and handed to the 16 row lanes through the strip
//     (32 registers less than the float4-per-lane form: that is what buys the third wave);
//   * what the slabs of a point share (neighbour index, coordinates, H / G2 / Wsm rows: 128 - 256 B per row) is re-read by each wave
//     (L1 / L2 hits); sums over all channels of a row (the geometry branch's g_t1n) are LINEAR in the slab's share, so each wave adds
//     its share to its own partial sums and the block's epilogue adds the waves in order -- nothing crosses waves inside the loop;
//   * one resident round of workgroups (3 per CU), 4 - 50 trips per wave: the prologue and the partial row are paid per workgroup.
// Same math, same partial-row layouts, same reducers as the one-point-per-wave passes (which stay for the open-form geometry backward
// and as the A/B baseline, PDFOPS_PT_SLAB=0).  Summation orders are fixed: bit-reproducible.  Measurements: profiles/r06_slab_b3_ab.txt.
#include "fused_layer_mfma.h"

namespace fls {

using namespace flm;

template <int C> __host__ __device__ constexpr int waves_of() { return C / 64 > 4 ? C / 64 : 4; }   // waves per workgroup

// the points of one (workgroup, point lane): PPB lanes interleave inside the workgroup's chunk of the visiting order
struct SlabWalk {
    long t, end, stride;
    const int *order;
    __device__ __forceinline__ SlabWalk(const LayerArgs &A, int plane, int ppb) {
        const unsigned g = gridDim.x;
        order = A.order;
        if (A.chunked & 1) {
            const unsigned blk = pdf_xcd_chunked_block(blockIdx.x, g);
            const long per = ((long)A.N + g - 1) / g;
            t = (long)blk * per + plane;
            end = (long)(blk + 1) * per < (long)A.N ? (long)(blk + 1) * per : (long)A.N;
            stride = ppb;
        } else {
            t = (long)blockIdx.x * ppb + plane;
            end = A.N;
            stride = (long)g * ppb;
        }
    }
    __device__ __forceinline__ bool valid() const { return t < end; }
    __device__ __forceinline__ bool has_next() const { return t + stride < end; }
    __device__ __forceinline__ long point() const { return order ? (long)order[t] : t; }
    __device__ __forceinline__ long next_point() const { return order ? (long)order[t + stride] : t + stride; }
    __device__ __forceinline__ void step() { t += stride; }
};

// ---- the wave's private LDS strip (floats): its 64 channels' constants
//   [s1 | t1 | backward: mean1 | rstd1 | s1 * sum g_y1 / rows | s1 * sum g_y1*rhat / rows | the trip's x_q | g_out |
//    vector form only: Wp2[:, 0] | Wp2[:, 1] | Wp2[:, 2] (a-major: packed fp32 math over channel pairs) | bp2]
constexpr int SC_S1 = 0, SC_T1 = 64, SC_M1 = 128, SC_R1 = 192, SC_SA = 256, SC_SB = 320, SC_XQ = 384, SC_GO = 448, SC_W = 512, SC_B = 704, SC_N = 768;   // SC_W: three 64-float rows Wp2[:, a]
constexpr int SC_N_MM = 512;   // (the matrix-core form keeps Wp2 / bp2 as register operands: no W / B rows)

// p_r of the lane's four channels of LOCAL group gl = 4 jj + kq (channels 4 gl .. 4 gl + 3 of the slab): three vector FMAs (v_pk_fma_f32)
__device__ __forceinline__ f32x4 pos4s(const float *sc, int gl, const float *t1n) {
    return t1n[0] * ld4(sc + SC_W + 4 * gl) + (t1n[1] * ld4(sc + SC_W + 64 + 4 * gl) + (t1n[2] * ld4(sc + SC_W + 128 + 4 * gl) + ld4(sc + SC_B + 4 * gl)));
}

// Every lane loads the constants of channel c0 + lane and stores them into the wave's strip (all loads first; no block barrier: the
// strip is private to the wave -- a wave_sync orders it)
template <int C, bool BWD, bool WB>
__device__ __forceinline__ void stage_slab_consts(float *sc, const LayerArgs &A, int c0, int lane, const float *S) {
    const int ch = c0 + lane;
    const float w0 = gp(A.Wp2)[3 * ch], w1 = gp(A.Wp2)[3 * ch + 1], w2 = gp(A.Wp2)[3 * ch + 2], b = gp(A.bp2)[ch], s1 = gp(A.s1)[ch], t1 = gp(A.t1)[ch];
    float m1 = 0.f, r1 = 0.f, sa = 0.f, sb = 0.f;
    if constexpr (BWD) { m1 = gp(A.mean)[3 + ch]; r1 = gp(A.rstd)[3 + ch]; sa = S[ch] * A.inv_rows; sb = S[C + ch] * A.inv_rows; }
    if constexpr (WB) { sc[SC_W + lane] = w0; sc[SC_W + 64 + lane] = w1; sc[SC_W + 128 + lane] = w2; sc[SC_B + lane] = b; }
    sc[SC_S1 + lane] = s1; sc[SC_T1 + lane] = t1;
    if constexpr (BWD) { sc[SC_M1 + lane] = m1; sc[SC_R1 + lane] = r1; sc[SC_SA + lane] = s1 * sa; sc[SC_SB + lane] = s1 * sb; }
}

// The strip's constants are invariant over the point loop: left alone, the compiler hoists all ~40 float4 reads per 16-channel block out
// of it (160 registers, one wave per SIMD less).  An opaque copy of the strip pointer per trip keeps the reads at their uses.
// (the OFFSET is made opaque, not the pointer: an opaque pointer loses its address space and every read becomes a flat load)
__device__ __forceinline__ int per_trip_zero() { int z = 0; asm volatile("" : "+v"(z)); return z; }

// ================================================================================================ B3
// partial row per workgroup: [sum g_yp (3) | sum g_yp*that (3) | pad 2 | g_bp2 (C) | g_Wp2 (C*3) | sum g_yp (x) rel (9) | pad 7]   (as flm::k_b3)
// Only with the closed-form geometry backward (A.mom != nullptr, fl::k_colsum's extra block): G3 is not written -- a wave holds its
// slab's share of g_t1n only, which is all the 15 sums need.
// constants | g_r tile | g_pr tile | [t1n | 1] of the 16 rows | the 15 geometry sums per row lane
template <bool MM> constexpr int b3_wave_floats() { return (MM ? SC_N_MM : SC_N) + 16 * TS + 16 * (MM ? 80 : TS) + 64 + 240; }
template <int C, bool MM> constexpr size_t b3_lds_floats() { return (size_t)7 * csp_of(C) + (size_t)waves_of<C>() * b3_wave_floats<MM>(); }

// The row data of one trip (one point, the wave's slab), as it comes out of global memory
template <int NOB>
struct B3Rows {
    float pn[3], pi[3];
    f32x4 hh[NOB], g2[NOB], w[NOB], xk[4];
    float xq, go;   // x_q / g_out of channel c0 + lane (row-invariant: one value per lane, handed to the row lanes through the strip)
};
template <int C, bool BF>
__device__ __forceinline__ void b3_load(B3Rows<nob_of(C)> &D, const LayerArgs &A, long i, int nb, int row, int kq, int c0) {
    constexpr int CS = C / 8, NOB = nob_of(C);
    const size_t nbc = (size_t)max(nb, 0), ri = (size_t)i * 16 + row;
#pragma unroll
    for (int b = 0; b < 3; ++b) { D.pn[b] = A.p[nbc * 3 + b]; D.pi[b] = A.p[(size_t)i * 3 + b]; }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        const size_t o = ri * CS + unit_off<C>(ob, kq);
        D.hh[ob] = ld_row4<BF>(A.H, o); D.g2[ob] = ld_row4<BF>(A.G2, o); D.w[ob] = ld_row4<BF>(A.Wsm, o);
    }
    const float *xkr = A.xk + nbc * C + c0 + 4 * kq;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) D.xk[jj] = ld4(xkr + 16 * jj);
    const int lane = 16 * kq + row;
    D.xq = A.xq[(size_t)i * C + c0 + lane]; D.go = A.gout[(size_t)i * C + c0 + lane];
}

// VAR (A/B builds, PDFOPS_FLS_VAR): 0 (default) = the elementwise part as packed fp32 vector code;  1 = the skinny products and the row
// reductions of the pass on the matrix cores as well --
//   p_r:     r^T = Wp2x . t1nx^T + (x_k - x_q)^T, one 16x16x4 product per 16-channel block with the gathered rows as the C operand
//            (Wp2x = [Wp2 | bp2], t1nx = [t1n | 1]: K = 4 is exactly one step);
//   g_t1n:   (Wp2^T g_pr^T): reduction over the channels, the D fragment's first three registers of the kq == 0 lanes ARE the row's g_t1n;
//   g_Wp2 / g_bp2: [t1n | 1]^T g_pr over the 16 rows, accumulated across the trips in four persistent D fragments.
// Measured SLOWER (36.9 vs 32.5 us at C = 256, 48.0 vs 46.3 at C = 128, profiles/r06_slab_b3_ab.txt): ~190 vector / LDS instructions per
// trip become 36 matrix instructions, but they are 36 more 32-cycle products on dependent accumulator chains of a pipe that already
// carries the trip's 32 - 64.  Kept as the A/B evidence.
template <int C, bool BF, int VAR>
__global__ __launch_bounds__(64 * waves_of<C>()) __attribute__((amdgpu_waves_per_eu(C == 512 ? 2 : 3, 8))) void k_b3(LayerArgs A) {
    constexpr int CS = C / 8, NOB = nob_of(C), CSP = csp_of(C), NSLAB = C / 64, WV = waves_of<C>(), PPB = WV / NSLAB, W = 8 + 4 * C + 16;
    constexpr bool MM = VAR == 1, PRM = MM;
    constexpr int T2 = MM ? 80 : TS;   // row stride of the g_pr tile (80: the lanes-along-channels reads of the B operand are conflict-free)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const int q = wv % NSLAB, plane = wv / NSLAB, c0 = 64 * q;
    float *ucst = lds;                                   // per-unit constants (stage_units), block-shared; sums = B1's
    constexpr int WF = b3_wave_floats<MM>(), SCN = MM ? SC_N_MM : SC_N;
    float *scw = lds + 7 * CSP + wv * WF;                // the wave's strip
    float *tile = scw + SCN, *tile2 = tile + 16 * TS, *t1nt = tile2 + 16 * T2;   // t1nt: [row][4] = [t1n (3) | 1]
    float *gacc = t1nt + 64 + row * 15;   // the row lane's share of the 15 sums of the geometry branch (lanes kq == 0; stride 15: conflict-free)
    SlabWalk pw(A, plane, PPB);
    // the first trip's neighbour index goes out before anything else: its round trip overlaps the staging below
    int nb_next = pw.valid() ? A.idx[pw.point() * 16 + row] : -1;
    stage_units<C, true>(ucst, A, gp(A.sums2));
    stage_slab_consts<C, true, !MM>(scw, A, c0, lane, gp(A.sums));
    // the slab of Ww1 as the A operand of (Ww1^T g_h): wf[ob][e][jj] = Ww1[16 ob + 4 kq + e][c0 + 16 jj + row]   (padding units: 0)
    float wf[NOB][4][4];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int u = 16 * ob + 4 * kq + e;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float v = gp(A.Ww1)[(size_t)min(u, CS - 1) * C + c0 + 16 * jj + row];
                wf[ob][e][jj] = u < CS ? v : 0.f;
            }
        }
    // Wp2 of the slab as matrix operands (lane (i = l & 15, k = l >> 4)):
    //   wpx[jj]    = [Wp2 | bp2][c0 + 16 jj + i][k]            A of r^T = Wp2x t1nx^T + C
    //   wpt[jj][e] = Wp2[c0 + 16 jj + 4 k + e][i], i < 3       A of g_t1n^T = Wp2^T g_pr^T   (rows 3 .. 15 of the block: zero)
    float wpx[PRM ? 4 : 1], wpt[MM ? 4 : 1][4];
    if constexpr (PRM) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int ch = c0 + 16 * jj + row;
            const float wv3 = gp(A.Wp2)[3 * ch + min(kq, 2)], bv = gp(A.bp2)[ch];
            wpx[jj] = kq < 3 ? wv3 : bv;
            if constexpr (MM) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = gp(A.Wp2)[3 * (c0 + 16 * jj + 4 * kq + e) + min(row, 2)];
                    wpt[jj][e] = row < 3 ? v : 0.f;
                }
            }
        }
        t1nt[lane] = 1.f;   // (slot 3 of every row stays 1: the bias column)
    }
    const GeoW G = geo_weights(A);
    if (kq == 0) {
#pragma unroll
        for (int e = 0; e < 15; ++e) gacc[e] = 0.f;
    }
    float sbp2 = 0.f, awp2[3] = {0.f, 0.f, 0.f};
    f32x4 pacc[4] = {zero4(), zero4(), zero4(), zero4()};   // MM: [g_Wp2[ch][0..2] | g_bp2[ch]] of channel c0 + 16 jj + l, lanes l < 16
    B3Rows<NOB> D;
    __syncthreads();
    for (; pw.valid(); pw.step()) {
        const long i = pw.point();
        const int nb = nb_next;
        const float *sc = scw + per_trip_zero();
        // ---------------- every global load of the trip
        b3_load<C, BF>(D, A, i, nb, row, kq, c0);
        nb_next = A.idx[(pw.has_next() ? pw.next_point() : i) * 16 + row];
        __builtin_amdgcn_sched_barrier(0);
        scw[SC_XQ + lane] = D.xq; scw[SC_GO + lane] = D.go;
        const Geo R = geo_of(G, nb, D.pn, D.pi);
        if (kq == 0) { t1nt[row * 4 + 0] = R.t1n[0]; t1nt[row * 4 + 1] = R.t1n[1]; t1nt[row * 4 + 2] = R.t1n[2]; }
        f32x4 gh[NOB], w[NOB];
        hidden_grad<C>(ucst, kq, D.hh, D.g2, gh);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) w[ob] = sel4(16 * ob + 4 * kq < CS, D.w[ob]);
        if (CS < 16) {   // channels of lanes kq = 2, 3 use the units of lanes kq - 2
            const f32x4 x = xchg32(w[0]);
            if (kq >= 2) w[0] = x;
        }
        f32x4 gt[3] = {zero4(), zero4(), zero4()};   // VAR 0: g_t1n as per-lane partial sums over the lane's channels
        f32x4 gtm = zero4();                          // MM: D fragment of Wp2^T g_pr^T
        const float live = nb >= 0 ? 1.f : 0.f;
        const float tb = kq == 0 ? R.t1n[0] : kq == 1 ? R.t1n[1] : kq == 2 ? R.t1n[2] : 1.f;   // B of the p_r product: [t1n | 1][row][k = kq]
        wave_sync();   // (x_q / g_out of the trip are in the strip)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int gl = 4 * jj + kq;   // local channel group: channels c0 + 4 gl ..+4
            f32x4 acc = zero4();
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ob][e][jj], gh[ob][e], acc, 0, 0, 0);
            f32x4 r, wa0, wa1, wa2;
            if constexpr (!MM) { wa0 = ld4(sc + SC_W + 4 * gl); wa1 = ld4(sc + SC_W + 64 + 4 * gl); wa2 = ld4(sc + SC_W + 128 + 4 * gl); }
            if constexpr (PRM) {
                r = __builtin_amdgcn_mfma_f32_16x16x4f32(wpx[jj], tb, D.xk[jj] * live - ld4(sc + SC_XQ + 4 * gl), 0, 0, 0);
            } else {
                const f32x4 pr = R.t1n[0] * wa0 + (R.t1n[1] * wa1 + (R.t1n[2] * wa2 + ld4(sc + SC_B + 4 * gl)));
                r = (D.xk[jj] * live - ld4(sc + SC_XQ + 4 * gl)) + pr;
            }
            const f32x4 s1 = ld4(sc + SC_S1 + 4 * gl);
            const f32x4 y1 = r * s1 + ld4(sc + SC_T1 + 4 * gl);
            f32x4 gy1;
#pragma unroll
            for (int e = 0; e < 4; ++e) gy1[e] = y1[e] > 0.f ? acc[e] : 0.f;
            // BN1 backward: g_r = s1 * (g_y1 - mean(g_y1) - rhat * mean(g_y1 * rhat)), the two means pre-multiplied by s1 in the strip
            const f32x4 rhat = (r - ld4(sc + SC_M1 + 4 * gl)) * ld4(sc + SC_R1 + 4 * gl);
            const f32x4 gr = (s1 * gy1 - ld4(sc + SC_SA + 4 * gl)) - rhat * ld4(sc + SC_SB + 4 * gl);
            st4(tile + row * TS + 16 * jj + 4 * kq, gr);
            const f32x4 gpr = gr + ld4(sc + SC_GO + 4 * gl) * w[jj % NOB];   // + the aggregation's share of p_r  ((4 q + jj) mod NOB = jj mod NOB: NOB divides 4)
            st4(tile2 + row * T2 + 16 * jj + 4 * kq, gpr);
            if constexpr (MM) {
#pragma unroll
                for (int e = 0; e < 4; ++e) gtm = __builtin_amdgcn_mfma_f32_16x16x4f32(wpt[jj][e], gpr[e], gtm, 0, 0, 0);
            } else {
                gt[0] += gpr * wa0; gt[1] += gpr * wa1; gt[2] += gpr * wa2;
            }
        }
        float gt1n[3];
        if constexpr (MM) {
            gt1n[0] = gtm[0]; gt1n[1] = gtm[1]; gt1n[2] = gtm[2];   // (lanes kq == 0: rows 0 .. 3 of the D fragment = a)
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float v = (gt[a][0] + gt[a][1]) + (gt[a][2] + gt[a][3]);
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                gt1n[a] = v;
            }
        }
        wave_sync();
        {   // lanes along channels: g_r rows out (256 B per row), g_xq[i] = - sum_rows g_r
            float acc = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const float v = tile[rr * TS + lane];
                acc += v;
                const size_t o = ((size_t)i * 16 + rr) * C + c0 + lane;   // g_xk = segmented sum of these rows
                if constexpr (BF) __builtin_nontemporal_store((unsigned short)fl::f2bf(v), reinterpret_cast<unsigned short *>(A.GR) + o);
                else __builtin_nontemporal_store(v, A.GR + o);
            }
            A.gxq[(size_t)i * C + c0 + lane] = -acc;
        }
        if constexpr (MM) {   // g_Wp2 / g_bp2: [t1n | 1]^T (16 rows) . g_pr (16 rows x 64 channels), four k-steps of four rows
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float a4 = t1nt[(4 * s4 + kq) * 4 + (row & 3)], av = row < 4 ? a4 : 0.f;   // A[i = l & 15][k = row 4 s4 + kq] = [t1n | 1 | 0 ..]
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    pacc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tile2[(4 * s4 + kq) * T2 + 16 * jj + row], pacc[jj], 0, 0, 0);
            }
        } else {   // g_bp2 / g_Wp2 of channel c0 + lane
            float cb2 = 0.f, cw2[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const float v = tile2[rr * T2 + lane];
                cb2 += v;
                cw2[0] += v * t1nt[rr * 4 + 0]; cw2[1] += v * t1nt[rr * 4 + 1]; cw2[2] += v * t1nt[rr * 4 + 2];
            }
            sbp2 += cb2; awp2[0] += cw2[0]; awp2[1] += cw2[1]; awp2[2] += cw2[2];
        }
        // the slab's share of g_t1n -> its share of the 15 sums of the geometry branch (ReLU mask of BNp; linear in the share)
        if (kq == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float v = R.t1n[a] > 0.f ? gt1n[a] : 0.f;
                gacc[a] += v;
                gacc[3 + a] += v * ((R.t1[a] - A.mean[a]) * A.rstd[a]);
                gacc[6 + 3 * a + 0] += v * R.rel[0]; gacc[6 + 3 * a + 1] += v * R.rel[1]; gacc[6 + 3 * a + 2] += v * R.rel[2];
            }
        }
        wave_sync();
    }
    // ---- the workgroup's partial row.  Columns of a slab: summed over the point lanes in lane order; the 15 geometry sums: over all
    // waves and their 16 row lanes, in that order.
    float *dst = A.partial + (size_t)blockIdx.x * W;
    float *crow = tile;   // 4 x 64 floats of the wave's own strip: [g_bp2 (64) | g_Wp2 (64 x 3)]
    if constexpr (MM) {
        if (lane < 16) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                crow[16 * jj + lane] = pacc[jj][3];
#pragma unroll
                for (int a = 0; a < 3; ++a) crow[64 + (16 * jj + lane) * 3 + a] = pacc[jj][a];
            }
        }
    } else {
        crow[lane] = sbp2;
#pragma unroll
        for (int a = 0; a < 3; ++a) crow[64 + lane * 3 + a] = awp2[a];
    }
    __syncthreads();
    if (plane == 0) {
        float b = 0.f, wsum[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int pl = 0; pl < PPB; ++pl) {
            const float *o = lds + 7 * CSP + (pl * NSLAB + q) * WF + SCN;
            b += o[lane]; wsum[0] += o[64 + lane * 3]; wsum[1] += o[64 + lane * 3 + 1]; wsum[2] += o[64 + lane * 3 + 2];
        }
        dst[8 + c0 + lane] = b;
#pragma unroll
        for (int a = 0; a < 3; ++a) dst[8 + C + 3 * (c0 + lane) + a] = wsum[a];
    }
    if (threadIdx.x < 24) {
        const int e = threadIdx.x;   // row slots [0, 8) and [8 + 4 C, 8 + 4 C + 16): sums 0..5, two pads, sums 6..14, seven pads
        const int src = e < 6 ? e : (e >= 8 && e < 17 ? e - 2 : -1);
        float v = 0.f;
        if (src >= 0) {
#pragma unroll 1
            for (int w = 0; w < WV; ++w) {
                const float *ga = lds + 7 * CSP + w * WF + SCN + 16 * TS + 16 * T2 + 64;
#pragma unroll
                for (int r = 0; r < 16; ++r) v += ga[r * 15 + src];
            }
        }
        dst[e < 8 ? e : 4 * C + e] = v;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
template <typename KernelT>
static void launch(KernelT kernel, dim3 grid, int threads, size_t lds_floats, const LayerArgs &A, hipStream_t s) {
    const size_t lds = lds_floats * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kernel<<<grid, threads, lds, s>>>(A);
}

// which widths take the slab form (PDFOPS_PT_SLAB: comma list of widths, "0" = none; default: all four)
bool enabled(int c) {
    static const unsigned mask = [] {
        const char *v = getenv("PDFOPS_PT_SLAB");
        if (!v) return 15u;
        unsigned m = 0;
        for (const char *p = v; *p;) {
            const int w = atoi(p);
            m |= w == 64 ? 1u : w == 128 ? 2u : w == 256 ? 4u : w == 512 ? 8u : 0u;
            while (*p && *p != ',') ++p;
            if (*p == ',') ++p;
        }
        return m;
    }();
    return (c == 64 && (mask & 1u)) || (c == 128 && (mask & 2u)) || (c == 256 && (mask & 4u)) || (c == 512 && (mask & 8u));
}

static inline int waves_of_rt(int c) { return c / 64 > 4 ? c / 64 : 4; }

// rows of the partial matrix = workgroups.  One resident round of workgroups (two per CU) with 6 - 50 trips per wave beats more, shorter
// ones at every level: 512 -> 35.8 us, 1,042 -> 55 us at 3,124 points x 256 channels; 99.6 vs 116 us at 50,000 x 64
// (profiles/r06_slab_b3_ab.txt) -- the prologue (constants, the slab of Ww1) and the partial row are paid per workgroup.
int b3_grid(long n, int c, int max_rows) {
    static const int env = [] { const char *v = getenv("PDFOPS_PT_SLAB_B3_GRID"); return v ? atoi(v) : 0; }();
    const int ppb = waves_of_rt(c) / (c / 64);
    long g = env > 0 ? env : (c == 512 ? 512 : 768);   // one resident round: 2 (C = 512: 8-wave workgroups, 226 registers) / 3 workgroups per CU
    if (g > (n + ppb - 1) / ppb) g = (n + ppb - 1) / ppb;
    if (g > max_rows) g = max_rows;
    if (g < 1) g = 1;
    return (int)g;
}

void launch_b3(const LayerArgs &A0, int c, int grid, hipStream_t s) {
    static const int var = [] { const char *v = getenv("PDFOPS_FLS_VAR"); return v ? atoi(v) : 0; }();   // code variants (A/B builds)
    const LayerArgs &A = A0;
#define K_(C_) do { if (A.bf16) launch(k_b3<C_, true, 0>, dim3(grid), 64 * waves_of<C_>(), b3_lds_floats<C_, false>(), A, s); \
                    else if (var == 1) launch(k_b3<C_, false, 1>, dim3(grid), 64 * waves_of<C_>(), b3_lds_floats<C_, true>(), A, s); \
                    else launch(k_b3<C_, false, 0>, dim3(grid), 64 * waves_of<C_>(), b3_lds_floats<C_, false>(), A, s); } while (0)
    if (c == 64) K_(64); else if (c == 128) K_(128); else if (c == 256) K_(256); else K_(512);
#undef K_
}

}  // namespace fls
