// Shared declarations of the fused PointTransformerLayer kernels (fused_layer.hip: row-per-lane passes for every supported
// shape; fused_layer_mfma.hip: matrix-core passes for nsample == 16, C >= 128).
#pragma once
#include "pdfops_common.h"

namespace fl {

constexpr int WPB = 4;          // waves per block

// Wave-uniform read-only operands (weights, BatchNorm coefficients, reduction results) are addressed through the
// constant address space: the compiler then fetches them with s_load into SGPRs instead of keeping one VGPR per value
// and lane (which spilled thousands of registers in the first version of these kernels).
typedef const float __attribute__((address_space(4))) *cfloat_p;
__host__ __device__ inline cfloat_p as_const(const float *p) { return (cfloat_p)(uintptr_t)p; }

struct LayerArgs {
    int N;                                   // points
    const float *xq, *xk, *xv, *p;           // (N,C) x3, (N,3)
    const int *idx;                          // (N,K)
    cfloat_p Wp1, bp1, Wp2, bp2;             // (3,3) (3) (C,3) (C)
    cfloat_p Ww1, bw1, Ww2, bw2;             // (CS,C) (CS) (CS,CS) (CS)
    cfloat_p sp, tp, s1, t1, s2, t2;         // BatchNorm scale/shift: y = x*s + t  (3,3,C,C,CS,CS)
    float *H;                                // (N,K,CS) pre-BN2 activations
    float *out;                              // (N,C)
    float *partial;                          // partial rows (one per workgroup) of the statistics / gradient sums of the current pass
    // ---- backward only
    const float *gout;                       // (N,C) gradient of the layer output
    cfloat_p mean, rstd;                     // saved batch statistics [p(3) | 1(C) | 2(CS)] (mean and rstd arrays)
    cfloat_p sums;                           // column sums of the previous backward pass (BatchNorm-backward terms)
    cfloat_p sums2;                          // B3 only: column sums of B1 (BN2-backward terms), `sums` then holds B2's
    float *G2, *G3;                          // (N,K,CS) grad wrt BN2 output (post-ReLU mask), (N,K,3) same for BNp
    float *gxq, *gxk, *gxv;                  // (N,C) gradients
    float *Wsm, *GR;                         // (N,K,CS) softmax weights (B1 -> g_xv gather), (N,K,C) g_r rows (B3 -> g_xk gather):
                                             // the scatters of g_xv / g_xk run as segmented gathers over the inverse kNN table
    float inv_rows;                          // 1 / (N*K)
    int bf16;                                // H / G2 / Wsm / GR hold bfloat16 (half the bytes of the same buffers) instead of fp32
    int chunked;                             // point walk of the matrix-core passes: 1 = contiguous chunk per workgroup (XCD-major), 0 = grid-stride
    // ---- forward, training: BatchNorm of the geometry branch from the relative-coordinate sums of the kNN table (geom_moments.hip)
    const double *mom;                       // 9 sums [S (3) | M (6)] over all (point, neighbour) rows, or nullptr (P1 + finalizer compute the statistics)
    cfloat_p gam_p, bet_p;                   // BNp gamma / beta (3 each)
    float *bnp_coef, *bnp_saved;             // where block 0 of P2 leaves [sp (3) | tp (3)] and mean (3) / rstd (3, at bnp_saved + bnp_T)
    float *bnp_rm, *bnp_rv;                  // running mean / var (nullable)
    int bnp_T;
    float bnp_eps, bnp_momentum;
    const int *order;                        // visiting order of the points (a permutation of 0 .. N-1, Morton order from the geometry
                                             // pre-pass) or nullptr: neighbouring points share neighbour rows -> per-XCD L2 hits
};

// (round 4) The row-per-lane passes read ~500 per-channel constants (weights, BatchNorm coefficients, column sums) through the constant
// address space.  They are loop-invariant, so the compiler hoisted all their scalar loads out of the tile loop -- and then had ~500 values
// to keep in ~100 scalar registers: fl::k_b3<32,8> spilled 617 SGPRs into VGPR lanes and restored them with ~1,000 v_readlane per tile
// (40 % of the pass's vector instructions; k_b2 494, k_p3 395 spills).  `fresh_consts` hands the tile loop's body pointers the optimiser
// cannot see through: the loads stay at their uses (merged into wide scalar loads that hit the scalar cache), live for a few instructions.
#ifdef PDF_NO_FRESH_CONSTS   // (A/B builds: the hoisted form of rounds 1-3)
__device__ __forceinline__ cfloat_p fresh(cfloat_p p) { return p; }
#else
__device__ __forceinline__ cfloat_p fresh(cfloat_p p) { asm volatile("" : "+s"(p)); return p; }
#endif
__device__ __forceinline__ LayerArgs fresh_consts(LayerArgs A) {
    A.Wp1 = fresh(A.Wp1); A.bp1 = fresh(A.bp1); A.Wp2 = fresh(A.Wp2); A.bp2 = fresh(A.bp2);
    A.Ww1 = fresh(A.Ww1); A.bw1 = fresh(A.bw1); A.Ww2 = fresh(A.Ww2); A.bw2 = fresh(A.bw2);
    A.sp = fresh(A.sp); A.tp = fresh(A.tp); A.s1 = fresh(A.s1); A.t1 = fresh(A.t1); A.s2 = fresh(A.s2); A.t2 = fresh(A.t2);
    A.mean = fresh(A.mean); A.rstd = fresh(A.rstd); A.sums = fresh(A.sums); A.sums2 = fresh(A.sums2);
    return A;
}

// The points one wave of the matrix-core passes visits: every workgroup owns one contiguous chunk of the visiting order (chunks dealt
// XCD-major, pdf_xcd_chunked_block), its WPB waves interleave inside the chunk.
struct PointWalk {
    long t, end;
    const int *order;
    long stride;
    __device__ __forceinline__ PointWalk(const LayerArgs &A, int wave_in_block) {
        const unsigned g = gridDim.x;
        order = A.order;
        if (A.chunked) {   // one contiguous chunk of the visiting order per workgroup, chunks dealt XCD-major
            const unsigned blk = pdf_xcd_chunked_block(blockIdx.x, g);
            const long per = ((long)A.N + g - 1) / g;
            t = (long)blk * per + wave_in_block;
            end = (long)(blk + 1) * per < (long)A.N ? (long)(blk + 1) * per : (long)A.N;
            stride = WPB;
        } else {           // grid-stride over the points, one point per wave and trip
            t = (long)blockIdx.x * WPB + wave_in_block;
            end = A.N;
            stride = (long)g * WPB;
        }
    }
    __device__ __forceinline__ bool valid() const { return t < end; }
    __device__ __forceinline__ bool has_next() const { return t + stride < end; }
    __device__ __forceinline__ long point() const { return order ? (long)order[t] : t; }
    __device__ __forceinline__ long next_point() const { return order ? (long)order[t + stride] : t + stride; }
    __device__ __forceinline__ void step() { t += stride; }
};

// ---- row arrays that only this layer writes and reads (H saved by the forward; G2, Wsm, GR scratch of the backward): stored as
// fp32 or, reduced-precision variant, as bfloat16 with round-to-nearest-even; all arithmetic stays fp32.  `idx` = element index.
__device__ __forceinline__ unsigned f2bf(float f) {
    const unsigned u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ void st_u4(float *base, size_t idx, float a, float b, float c, float d, int bf16) {
    if (bf16) {
        uint2 v; v.x = f2bf(a) | (f2bf(b) << 16); v.y = f2bf(c) | (f2bf(d) << 16);
        *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(base) + idx) = v;
    } else {
        *reinterpret_cast<float4 *>(base + idx) = make_float4(a, b, c, d);
    }
}
__device__ __forceinline__ float4 ld_u4(const float *base, size_t idx, int bf16) {
    if (bf16) {
        const uint2 v = *reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(base) + idx);
        return make_float4(bf2f(v.x & 0xffffu), bf2f(v.x >> 16), bf2f(v.y & 0xffffu), bf2f(v.y >> 16));
    }
    return *reinterpret_cast<const float4 *>(base + idx);
}
__device__ __forceinline__ void st_u1_stream(float *base, size_t idx, float v, int bf16) {
    if (bf16) __builtin_nontemporal_store((unsigned short)f2bf(v), reinterpret_cast<unsigned short *>(base) + idx);
    else __builtin_nontemporal_store(v, base + idx);
}

// scale / shift of the geometry branch's BatchNorm: from the moments (every block computes its own copy; `writer` also stores the
// coefficients, the saved statistics and the running-statistics update where the later passes and the backward read them), or as the
// finalizer left them.  t1 = W rel + b  ->  mean = W S / rows + b,  E[t1^2] = (W M W^T + 2 b W S) / rows + b^2.
struct BnP { float sp[3], tp[3]; };
__device__ __forceinline__ BnP bnp_of(const LayerArgs &A, long rows, bool writer) {
    BnP B;
    if (A.mom == nullptr) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { B.sp[a] = A.sp[a]; B.tp[a] = A.tp[a]; }
        return B;
    }
    const double S[3] = {A.mom[0], A.mom[1], A.mom[2]};
    const double M[3][3] = {{A.mom[3], A.mom[4], A.mom[5]}, {A.mom[4], A.mom[6], A.mom[7]}, {A.mom[5], A.mom[7], A.mom[8]}};
    const double inv = 1.0 / (double)rows;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double w[3] = {(double)A.Wp1[a * 3], (double)A.Wp1[a * 3 + 1], (double)A.Wp1[a * 3 + 2]}, b = (double)A.bp1[a];
        const double ws = w[0] * S[0] + w[1] * S[1] + w[2] * S[2];
        double wmw = 0.0;
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y) wmw += w[x] * M[x][y] * w[y];
        const double mean = ws * inv + b;
        double var = (wmw + 2.0 * b * ws) * inv + b * b - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)A.bnp_eps));
        B.sp[a] = A.gam_p[a] * rstd;
        B.tp[a] = A.bet_p[a] - (float)mean * B.sp[a];
        if (writer) {
            A.bnp_coef[a] = B.sp[a]; A.bnp_coef[3 + a] = B.tp[a];
            A.bnp_saved[a] = (float)mean; A.bnp_saved[A.bnp_T + a] = rstd;
            if (A.bnp_rm) {
                const double unbiased = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
                A.bnp_rm[a] = (1.f - A.bnp_momentum) * A.bnp_rm[a] + A.bnp_momentum * (float)mean;
                A.bnp_rv[a] = (1.f - A.bnp_momentum) * A.bnp_rv[a] + A.bnp_momentum * (float)unbiased;
            }
        }
    }
    return B;
}

// ---- partial rows.  Every pass accumulates its BatchNorm sums / parameter gradients per wave in registers; the WPB waves of a block
// then combine them in LDS (fixed order: deterministic) and the block stores ONE row.  The reducers (k_bn_finalize / k_colsum, ~280
// launches per step, pure latency) read `grid` rows instead of `grid * WPB`, and the wide weight-gradient rows cost a quarter of the
// traffic.  `o[col] = v` inside `emit` writes the wave's value of column `col`.
struct RowRef {
    float *p; bool add;
    __device__ __forceinline__ void operator=(float v) const { *p = add ? *p + v : v; }
};
struct RowAcc {
    float *row; bool add;
    __device__ __forceinline__ RowRef operator[](size_t i) const { return RowRef{row + i, add}; }
};
// `stage`: WPB * width floats of LDS the pass no longer needs.  All waves write their rows at once, then every column is summed in wave
// order into row 0.
template <typename F>
__device__ __forceinline__ void block_row(float *stage, int width, F &&emit) {
    __syncthreads();
    emit(RowAcc{stage + (threadIdx.x >> 6) * width, false});
    __syncthreads();
    for (int e = threadIdx.x; e < width; e += 64 * WPB) {
        float v = stage[e];
#pragma unroll
        for (int w = 1; w < WPB; ++w) v += stage[w * width + e];
        stage[e] = v;
    }
    __syncthreads();
}
// `stage`: width floats only (rows too wide for WPB copies): wave 0 stores, waves 1.. add, one after the other.
template <typename F>
__device__ __forceinline__ void block_row_seq(float *stage, F &&emit) {
    const int wv = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll 1
    for (int w = 0; w < WPB; ++w) {
        if (wv == w) emit(RowAcc{stage, w > 0});
        __syncthreads();
    }
}
__device__ __forceinline__ void store_row(const float *stage, int width, float *dst) {
    for (int e = threadIdx.x; e < width; e += 64 * WPB) dst[e] = stage[e];
}

}  // namespace fl

namespace flm {
// matrix-core variants (fused_layer_mfma.hip); `grid` blocks of 64 * WPB threads, partial rows = grid
bool supported(int nsample, int c);
void launch_p2(const fl::LayerArgs &A, int c, int grid, hipStream_t s);
void launch_p3(const fl::LayerArgs &A, int c, bool stats, int grid, hipStream_t s);
int p4_grid(long n, int grid);   // blocks (= partial rows with statistics) of launch_p4
void launch_p4(const fl::LayerArgs &A, int c, int grid, bool stats, hipStream_t s);
void launch_b1(const fl::LayerArgs &A, int c, int grid, hipStream_t s);
void launch_b2(const fl::LayerArgs &A, int c, int grid, hipStream_t s);
void launch_b3(const fl::LayerArgs &A, int c, int grid, hipStream_t s);
bool supported_l1(int nsample, int c);   // level 1 (C = 32, nsample 8): two points per wave
void launch_b2_l1(const fl::LayerArgs &A, int grid, hipStream_t s);
void launch_b3_l1(const fl::LayerArgs &A, int grid, hipStream_t s);
}  // namespace flm

namespace fls {
// slab form (fused_layer_slab.hip): one wave = one 64-channel slab of one point; workgroups of max(4, C / 64) waves, partial rows = grid
bool enabled(int c);                                  // PDFOPS_PT_SLAB (default: C = 256, 512)
int b3_grid(long n, int c, int max_rows);
void launch_b3(const fl::LayerArgs &A, int c, int grid, hipStream_t s);   // closed-form geometry backward only (G3 is not written)
}  // namespace fls
