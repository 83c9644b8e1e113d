// Fused PointTransformerLayer passes, SLAB form (round 6): one wave = one 64-channel slab of one point.
//
// fused_layer_mfma.hip gives a wave ONE point and walks the point's C / 64 chunks one after the other: at levels 4 - 5 (3,124 / 780
// points, C = 256 / 512) that is one or two points per wave, each a serial chain of 4 - 8 chunks behind a block-wide staging of the
// C x C/8 weights -- ~50 us per pass whatever N is (profiles/r05_z_kernel_trace_stats.txt), two waves per SIMD, half of a wave's
// instructions LDS traffic (weight operands, transposition tiles) and column phases.  Here
//   * a workgroup's waves are the SLABS of one point stream (C >= 256: wave w = slab w; C = 128: two point lanes x two slabs; C = 64:
//     four point lanes): the dependent chain of a point shrinks C/64-fold and every SIMD holds 3 - 4 such waves;
//   * a wave keeps ITS slab of Ww1 in registers for the whole launch (CSP x 64 values = 16 NOB registers: no LDS copy of the weights,
//     no block barrier in the point loop, no sweeps at C = 512) and its 64 channels' constants in a private LDS strip;
//   * what the slabs of a point share (neighbour index, coordinates, H / G2 / Wsm rows: 128 - 256 B per row) is re-read by each wave
//     (L1 / L2 hits); sums over all channels of a row (the geometry branch's g_t1n) are LINEAR in the slab's share, so each wave adds
//     its share to its own partial sums and the block's epilogue adds the waves in order -- nothing crosses waves inside the loop.
// Same math, same partial-row layouts, same reducers as the one-point-per-wave passes (which stay: C = 64 / 128 defaults, the
// forward's eval path, the open-form geometry backward).  Summation orders are fixed: bit-reproducible.
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
        if (A.chunked) {
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
//   [Wp2 (64 x 3, channel-major) | bp2 | s1 | t1 | backward: mean1 | rstd1 | sum g_y1 / rows | sum g_y1*rhat / rows]
constexpr int SC_W = 0, SC_B = 192, SC_S1 = 256, SC_T1 = 320, SC_M1 = 384, SC_R1 = 448, SC_SA = 512, SC_SB = 576, SC_N = 640;

// p_r of the lane's four channels of LOCAL group gl = 4 jj + kq (channels 4 gl .. 4 gl + 3 of the slab)
__device__ __forceinline__ f32x4 pos4s(const float *sc, int gl, const float *t1n) {
    const f32x4 w0 = ld4(sc + SC_W + 12 * gl), w1 = ld4(sc + SC_W + 12 * gl + 4), w2 = ld4(sc + SC_W + 12 * gl + 8), b = ld4(sc + SC_B + 4 * gl);
    const float w[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0], w2[1], w2[2], w2[3]};
    f32x4 pr;
#pragma unroll
    for (int e = 0; e < 4; ++e) pr[e] = t1n[0] * w[3 * e] + t1n[1] * w[3 * e + 1] + t1n[2] * w[3 * e + 2] + b[e];
    return pr;
}

// Every lane loads the constants of channel c0 + lane and stores them into the wave's strip (all loads first; no block barrier: the
// strip is private to the wave -- a wave_sync orders it)
template <int C, bool BWD>
__device__ __forceinline__ void stage_slab_consts(float *sc, const LayerArgs &A, int c0, int lane, const float *S) {
    const int ch = c0 + lane;
    const float w0 = gp(A.Wp2)[3 * ch], w1 = gp(A.Wp2)[3 * ch + 1], w2 = gp(A.Wp2)[3 * ch + 2], b = gp(A.bp2)[ch], s1 = gp(A.s1)[ch], t1 = gp(A.t1)[ch];
    float m1 = 0.f, r1 = 0.f, sa = 0.f, sb = 0.f;
    if constexpr (BWD) { m1 = gp(A.mean)[3 + ch]; r1 = gp(A.rstd)[3 + ch]; sa = S[ch] * A.inv_rows; sb = S[C + ch] * A.inv_rows; }
    sc[SC_W + 3 * lane] = w0; sc[SC_W + 3 * lane + 1] = w1; sc[SC_W + 3 * lane + 2] = w2;
    sc[SC_B + lane] = b; sc[SC_S1 + lane] = s1; sc[SC_T1 + lane] = t1;
    if constexpr (BWD) { sc[SC_M1 + lane] = m1; sc[SC_R1 + lane] = r1; sc[SC_SA + lane] = sa; sc[SC_SB + lane] = sb; }
}

// The strip's constants are invariant over the point loop: left alone, the compiler hoists all ~40 float4 reads per 16-channel block out
// of it (160 registers, one wave per SIMD less).  An opaque copy of the strip pointer per trip keeps the reads at their uses.
__device__ __forceinline__ const float *per_trip(const float *p) { asm volatile("" : "+v"(p)); return p; }

// ================================================================================================ B3
// partial row per workgroup: [sum g_yp (3) | sum g_yp*that (3) | pad 2 | g_bp2 (C) | g_Wp2 (C*3) | sum g_yp (x) rel (9) | pad 7]   (as flm::k_b3)
// Only with the closed-form geometry backward (A.mom != nullptr, fl::k_colsum's extra block): G3 is not written -- a wave holds its
// slab's share of g_t1n only, which is all the 15 sums need.
constexpr int B3_WAVE_FLOATS = SC_N + 32 * TS + 64;   // constants | g_r tile | g_pr tile | t1n of the 16 rows
template <int C> constexpr size_t b3_lds_floats() { return (size_t)7 * csp_of(C) + (size_t)waves_of<C>() * B3_WAVE_FLOATS; }

template <int C, bool BF>
__global__ __launch_bounds__(64 * waves_of<C>()) void k_b3(LayerArgs A) {
    constexpr int CS = C / 8, NOB = nob_of(C), CSP = csp_of(C), NSLAB = C / 64, WV = waves_of<C>(), PPB = WV / NSLAB, W = 8 + 4 * C + 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const int q = wv % NSLAB, plane = wv / NSLAB, c0 = 64 * q;
    float *ucst = lds;                                   // per-unit constants (stage_units), block-shared; sums = B1's
    float *scw = lds + 7 * CSP + wv * B3_WAVE_FLOATS;    // the wave's strip
    float *tile = scw + SC_N, *tile2 = tile + 16 * TS, *t1nt = tile2 + 16 * TS;
    stage_units<C, true>(ucst, A, gp(A.sums2));
    stage_slab_consts<C, true>(scw, A, c0, lane, gp(A.sums));
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
    const GeoW G = geo_weights(A);
    float greg[15];
#pragma unroll
    for (int e = 0; e < 15; ++e) greg[e] = 0.f;
    float sbp2 = 0.f, awp2[3] = {0.f, 0.f, 0.f};
    __syncthreads();
    SlabWalk pw(A, plane, PPB);
    int nb_next = pw.valid() ? A.idx[pw.point() * 16 + row] : -1;
    for (; pw.valid(); pw.step()) {
        const long i = pw.point();
        const size_t ri = (size_t)i * 16 + row;
        const int nb = nb_next;
        const size_t nbc = (size_t)max(nb, 0);
        const float *sc = per_trip(scw);
        // ---------------- every global load of the trip
        nb_next = A.idx[(pw.has_next() ? pw.next_point() : i) * 16 + row];
        float pn[3], pi[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) { pn[b] = A.p[nbc * 3 + b]; pi[b] = A.p[(size_t)i * 3 + b]; }
        f32x4 hh[NOB], g2[NOB], w[NOB], xk[4], xq[4], go[4];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const size_t o = ri * CS + unit_off<C>(ob, kq);
            hh[ob] = ld_row4<BF>(A.H, o); g2[ob] = ld_row4<BF>(A.G2, o); w[ob] = ld_row4<BF>(A.Wsm, o);
        }
        const float *xkr = A.xk + nbc * C + c0 + 4 * kq, *xqr = A.xq + (size_t)i * C + c0 + 4 * kq, *gor = A.gout + (size_t)i * C + c0 + 4 * kq;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { xk[jj] = ld4(xkr + 16 * jj); xq[jj] = ld4(xqr + 16 * jj); go[jj] = ld4(gor + 16 * jj); }
        __builtin_amdgcn_sched_barrier(0);
        const Geo R = geo_of(G, nb, pn, pi);
        if (kq == 0) { t1nt[row * 4 + 0] = R.t1n[0]; t1nt[row * 4 + 1] = R.t1n[1]; t1nt[row * 4 + 2] = R.t1n[2]; }
        f32x4 gh[NOB];
        hidden_grad<C>(ucst, kq, hh, g2, gh);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) w[ob] = sel4(16 * ob + 4 * kq < CS, w[ob]);
        if (CS < 16) {   // channels of lanes kq = 2, 3 use the units of lanes kq - 2
            const f32x4 x = xchg32(w[0]);
            if (kq >= 2) w[0] = x;
        }
        float gt1n[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int gl = 4 * jj + kq;   // local channel group: channels c0 + 4 gl ..+4
            f32x4 acc = zero4();
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ob][e][jj], gh[ob][e], acc, 0, 0, 0);
            const f32x4 r = (sel4(nb >= 0, xk[jj]) - xq[jj]) + pos4s(sc, gl, R.t1n);
            const f32x4 s1 = ld4(sc + SC_S1 + 4 * gl);
            const f32x4 y1 = r * s1 + ld4(sc + SC_T1 + 4 * gl);
            f32x4 gy1;
#pragma unroll
            for (int e = 0; e < 4; ++e) gy1[e] = y1[e] > 0.f ? acc[e] : 0.f;
            // BN1 backward: g_r = s1 * (g_y1 - mean(g_y1) - rhat * mean(g_y1 * rhat))
            const f32x4 rhat = (r - ld4(sc + SC_M1 + 4 * gl)) * ld4(sc + SC_R1 + 4 * gl);
            const f32x4 gr = s1 * (gy1 - ld4(sc + SC_SA + 4 * gl) - rhat * ld4(sc + SC_SB + 4 * gl));
            st4(tile + row * TS + 16 * jj + 4 * kq, gr);
            const f32x4 gpr = gr + go[jj] * w[jj % NOB];   // + the aggregation's share of p_r  ((4 q + jj) mod NOB = jj mod NOB: NOB divides 4)
            st4(tile2 + row * TS + 16 * jj + 4 * kq, gpr);
            const f32x4 w0 = ld4(sc + SC_W + 12 * gl), w1 = ld4(sc + SC_W + 12 * gl + 4), w2v = ld4(sc + SC_W + 12 * gl + 8);
            const float wp[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2v[0], w2v[1], w2v[2], w2v[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gt1n[0] += gpr[e] * wp[3 * e]; gt1n[1] += gpr[e] * wp[3 * e + 1]; gt1n[2] += gpr[e] * wp[3 * e + 2];
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
        {   // g_bp2 / g_Wp2 of channel c0 + lane
            float cb2 = 0.f, cw2[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const float v = tile2[rr * TS + lane];
                cb2 += v;
                cw2[0] += v * t1nt[rr * 4 + 0]; cw2[1] += v * t1nt[rr * 4 + 1]; cw2[2] += v * t1nt[rr * 4 + 2];
            }
            sbp2 += cb2; awp2[0] += cw2[0]; awp2[1] += cw2[1]; awp2[2] += cw2[2];
        }
        // the slab's share of g_t1n -> its share of the 15 sums of the geometry branch (ReLU mask of BNp; linear in the share)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = gt1n[a];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0) {
                v = R.t1n[a] > 0.f ? v : 0.f;
                greg[a] += v;
                greg[3 + a] += v * ((R.t1[a] - A.mean[a]) * A.rstd[a]);
                greg[6 + 3 * a + 0] += v * R.rel[0]; greg[6 + 3 * a + 1] += v * R.rel[1]; greg[6 + 3 * a + 2] += v * R.rel[2];
            }
        }
        wave_sync();
    }
    // ---- the workgroup's partial row.  Columns of a slab: summed over the point lanes in lane order; the 15 geometry sums: over all waves.
    float *dst = A.partial + (size_t)blockIdx.x * W;
    float *crow = tile;   // 4 x 64 + 16 floats of the wave's own strip
    crow[lane] = sbp2;
#pragma unroll
    for (int a = 0; a < 3; ++a) crow[64 + lane * 3 + a] = awp2[a];
#pragma unroll
    for (int e = 0; e < 15; ++e) {
        const float x = pdf_wave_sum_f32(kq == 0 ? greg[e] : 0.f);
        if (lane == 0) crow[256 + e] = x;
    }
    __syncthreads();
    if (plane == 0) {
        float b = 0.f, wsum[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int pl = 0; pl < PPB; ++pl) {
            const float *o = lds + 7 * CSP + (pl * NSLAB + q) * B3_WAVE_FLOATS + SC_N;
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
#pragma unroll
            for (int w = 0; w < WV; ++w) v += (lds + 7 * CSP + w * B3_WAVE_FLOATS + SC_N)[256 + src];
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

// which widths take the slab form (PDFOPS_PT_SLAB: comma list of widths, "0" = none; default: 256,512)
bool enabled(int c) {
    static const unsigned mask = [] {
        const char *v = getenv("PDFOPS_PT_SLAB");
        if (!v) return (1u << 2) | (1u << 3);
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

// rows of the partial matrix = workgroups: every wave should see ~2 points (one to prefetch behind), bounded by the scratch (`max_rows`)
int b3_grid(long n, int c, int max_rows) {
    static const int env = [] { const char *v = getenv("PDFOPS_PT_SLAB_B3_GRID"); return v ? atoi(v) : 0; }();
    const int ppb = waves_of_rt(c) / (c / 64);
    long g = env > 0 ? env : (n + 2L * ppb - 1) / (2L * ppb);
    if (g > 2048) g = 2048;
    if (g > max_rows) g = max_rows;
    if (g < 1) g = 1;
    return (int)g;
}

void launch_b3(const LayerArgs &A, int c, int grid, hipStream_t s) {
#define K_(C_) do { if (A.bf16) launch(k_b3<C_, true>, dim3(grid), 64 * waves_of<C_>(), b3_lds_floats<C_>(), A, s); \
                    else launch(k_b3<C_, false>, dim3(grid), 64 * waves_of<C_>(), b3_lds_floats<C_>(), A, s); } while (0)
    if (c == 64) K_(64); else if (c == 128) K_(128); else if (c == 256) K_(256); else K_(512);
#undef K_
}

}  // namespace fls
