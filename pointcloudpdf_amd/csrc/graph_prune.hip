// The graph stage of the PDF pseudo-label pass (pointcept/recognizers/ours/pointpdf_v1m1_base.py:309-380) as two single-workgroup
// kernels per scene.  Upstream moves the region's neighbour graph to the host and calls scipy.sparse.csgraph.minimum_spanning_tree,
// sklearn.mixture.GaussianMixture(n_components=2) on the tree's weights and scipy.sparse.csgraph.connected_components on the weak
// tree edges.  A region holds a few thousand points and <= 64 neighbours each: the work is tens of thousands of edges with a
// round structure (Boruvka: <= log2(nodes) rounds; EM: <= 200 iterations), i.e. latency of dependent steps, not bandwidth -- one
// workgroup per scene walks the edge list once per round out of L2 and synchronises with workgroup barriers instead of kernel
// boundaries and host reads (the torch-op form of the same algorithms: 12 ms per scene, of which 6 ms a numpy EM on the host).
//
// k_forest: minimum spanning forest under the strict total order (weight, entry index) -- with a strict order the forest is unique, so
// the chosen entries ARE scipy's tree whenever the weights are distinct -- and, as a by-product, the component label of every listed
// node (the root it ended under).  With w == NULL every active edge is equal: the labels are plain connected components.
// k_gmm2: two-component 1-D mixture by EM in double: quartile start, 2-means to its fixed point, then sklearn's loop (E-step, M-step, stop
// when the mean log-likelihood moved < tol) -- with sklearn's defaults (tol 1e-3, 100 iterations) the fit lands where GaussianMixture(2)
// does (its loose tolerance stops ~5 steps after the k-means start; a fully converged EM ends at another cut: tests/test_pseudo_label.py).
#include "pdfops_common.h"
#include <algorithm>

namespace gp {

using u64 = unsigned long long;
constexpr int T = 1024;
constexpr int U = 8;   // entries of the edge walk in flight per thread
constexpr u64 NONE = ~0ull;

__device__ inline unsigned order_bits(float f) {   // float -> unsigned with the same order (-0 == +0)
    const unsigned b = __float_as_uint(f + 0.0f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// Atomics are performed in L2; a plain load of the same address may be served from this CU's vector cache: read what atomics wrote
// with an agent-scope load.
__device__ inline u64 load_l2(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifdef GP_PROFILE   // tools/probes/forest_phase_probe.py: wall-clock ticks (100 MHz) at the phase boundaries of every round, thread 0
__device__ long long gp_prof[64 * 8];
#define GP_TICK(round, slot) do { if (threadIdx.x == 0) gp_prof[(round) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define GP_TICK(round, slot) do { } while (0)
#endif

// One workgroup.  u, v: E directed entries (node ids < n); w: E weights or NULL; active: E flags or NULL (all); nodes: n_nodes ids
// (repeats allowed) covering every endpoint of an active entry.  Inside, a node is its SLOT in `nodes` (slot[x] = a position of x in
// the list; a repeat's other positions stay isolated phantom slots): su / sv (E ints of scratch) hold the entries' endpoints as slots
// (-1: inactive), and the two arrays every entry of every round touches -- the component of a slot and the running minimum of a
// component -- live in LDS when the list fits (LDS_NODES): a round's walk is then coalesced reads of su / sv plus LDS traffic.
// (With those two arrays in global memory one CU retires ~0.5 G atomics/s: 160 us for the 75k atomics of a 37k-entry first round.)
// On return comp_out[x] = the ROOT of x's component for every listed x (the id of one of the component's nodes, the same for all of
// them); chosen (E bytes, or NULL): 1 for the entries of the forest.
constexpr int LDS_NODES = 12288;   // 12 B per slot: 144 KB of the CU's 160 KB

// (round 5) dev: [n_nodes, E] in device memory (the region's sizes, left there by pdf_region_edges) or NULL; the by-value E / n_nodes are then
// the CAPACITIES of the arrays.  `when`: 0 = always, 1 = only when the list fits the LDS form, 2 = only when it does not (the host cannot
// know, so it launches both forms and one of them returns at once).
template <bool LDS>
__device__ __forceinline__ void forest_body(long long n, int E, const long long *__restrict__ u, const long long *__restrict__ v,
                                            const float *__restrict__ w, const unsigned char *__restrict__ active,
                                            const long long *__restrict__ nodes, int n_nodes, int *comp_out, int *slot, int *su, int *sv,
                                            u64 *gbest, int *gcomp, int *parent, int *rootof, unsigned char *chosen,
                                            const int *__restrict__ dev, int when) {
    extern __shared__ u64 lds_dyn[];
    __shared__ int live;
    if (dev) {
        n_nodes = min(n_nodes, dev[0]);
        E = min(E, dev[1]);
        if ((when == 1 && n_nodes > LDS_NODES) || (when == 2 && n_nodes <= LDS_NODES) || n_nodes <= 0) return;
    }
    u64 *best;
    int *comp;
    if constexpr (LDS) {
        best = lds_dyn;
        comp = reinterpret_cast<int *>(lds_dyn + n_nodes);
    } else {
        best = gbest;
        comp = gcomp;
    }
    auto best_now = [&](int c) -> u64 {
        if constexpr (LDS) return best[c];
        else return load_l2(&best[c]);
    };
    const int t = threadIdx.x;
    for (int i = t; i < n_nodes; i += T) {
        const long long x = nodes[i];
        if (x >= 0 && x < n) slot[x] = i;   // (a repeated id keeps one of its positions: whichever store lands last)
        comp[i] = i;
    }
    if (t == 0) live = 0;
    __syncthreads();
    // `slot` is scratch that only the listed ids initialised: an endpoint outside [0, n) or missing from the list (a caller's mistake)
    // reads garbage there -- such an entry is dropped (its slot would not point back at the id), it must not index LDS
    auto slot_of = [&](long long x) -> int {
        if (x < 0 || x >= n) return -1;
        const int k = slot[x];
        return ((unsigned)k < (unsigned)n_nodes && nodes[k] == x) ? k : -1;
    };
    for (int e = t; e < E; e += T) {
        int a = -1, b = -1;
        if (!active || active[e]) {
            a = slot_of(u[e]);
            b = slot_of(v[e]);
        }
        const bool ok = a >= 0 && b >= 0;
        su[e] = ok ? a : -1;
        sv[e] = ok ? b : -1;
        if (chosen) chosen[e] = 0;
    }
    __syncthreads();
    for (int round = 0; round < 48; ++round) {   // (every round at least halves the number of components that still have an edge out)
        GP_TICK(round, 0);
        for (int i = t; i < n_nodes; i += T) best[i] = NONE;
        __syncthreads();
        GP_TICK(round, 1);
        int any = 0;
        // U entries per trip: their endpoint loads, then their component loads, are in flight together
        for (int e0 = t; e0 < E; e0 += T * U) {
            int a[U], b[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int e = e0 + j * T;
                a[j] = e < E ? su[e] : -1;
                b[j] = e < E ? sv[e] : -1;
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int x = a[j], y = b[j];
                a[j] = x >= 0 ? comp[x] : 0;
                b[j] = x >= 0 ? comp[y] : 0;
            }
            // An atomic only where the entry would lower the component's minimum as it stands: once components are large, thousands of
            // entries aim at the same few words and same-address atomics serialise.  The minimum only falls within a round, so a value
            // read earlier can only let a useless atomic through.
            u64 ma[U], mb[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const bool lv = a[j] != b[j];
                ma[j] = lv ? best_now(a[j]) : 0ull;
                mb[j] = lv ? best_now(b[j]) : 0ull;
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (a[j] == b[j]) continue;
                const int e = e0 + j * T;
                const u64 key = ((u64)(w ? order_bits(w[e]) : 0u) << 32) | (unsigned)e;
                if (key < ma[j]) atomicMin(&best[a[j]], key);
                if (key < mb[j]) atomicMin(&best[b[j]], key);
                any = 1;
            }
        }
        if (any) live = 1;
        __syncthreads();
        GP_TICK(round, 2);
        if (!live) break;
        // every component hooks onto the component at the other end of its lightest outgoing entry
        for (int c = t; c < n_nodes; c += T) {
            if (comp[c] != c) continue;
            const u64 bm = best_now(c);
            int p = c;
            if (bm != NONE) {
                const int e = (int)(unsigned)(bm & 0xffffffffull);
                if (chosen) chosen[e] = 1;
                const int x = comp[su[e]], y = comp[sv[e]];
                p = (x == c) ? y : x;
            }
            parent[c] = p;
        }
        __syncthreads();
        if (t == 0) live = 0;
        // two components that picked the same entry point at each other: the smaller slot stays a root
        for (int c = t; c < n_nodes; c += T) {
            if (comp[c] != c) continue;
            const int p = parent[c];
            if (p != c && c < p && parent[p] == c) parent[c] = c;
        }
        __syncthreads();
        for (int c = t; c < n_nodes; c += T) {
            if (comp[c] != c) continue;
            int r = parent[c];
            while (parent[r] != r) r = parent[r];
            rootof[c] = r;
        }
        __syncthreads();
        for (int i = t; i < n_nodes; i += T) comp[i] = rootof[comp[i]];   // (comp[i] is a root of the round before; a new root maps to itself)
        __syncthreads();
        GP_TICK(round, 3);
    }
    for (int i = t; i < n_nodes; i += T) {
        const long long x = nodes[i];
        if (x >= 0 && x < n) comp_out[x] = (int)nodes[comp[slot[x]]];
    }
}

template <bool LDS>
__global__ __launch_bounds__(T) void k_forest(long long n, int E, const long long *__restrict__ u, const long long *__restrict__ v,
                                              const float *__restrict__ w, const unsigned char *__restrict__ active,
                                              const long long *__restrict__ nodes, int n_nodes, int *comp_out, int *slot, int *su, int *sv,
                                              u64 *gbest, int *gcomp, int *parent, int *rootof, unsigned char *chosen,
                                              const int *__restrict__ dev = nullptr, int when = 0) {
    forest_body<LDS>(n, E, u, v, w, active, nodes, n_nodes, comp_out, slot, su, sv, gbest, gcomp, parent, rootof, chosen, dev, when);
}

// The forests of several scenes in ONE launch (grid = scenes; round 5: one launch per scene ran the scenes one after the other, 0.2 ms each).
// Scene s: points [start[s], start[s] + size[s]); its entry arrays hold size * stride entries at start * stride; its sizes are
// dev[s * dev_stride + 0 / 1]; workspace: the scenes' pdf_graph_forest_workspace_bytes(size, size * stride, size), 8-byte aligned, back to back.
constexpr int FOREST_BATCH = 16;
struct ForestBatch {
    int start[FOREST_BATCH], size[FOREST_BATCH];
    long ws_off[FOREST_BATCH];   // bytes
    int stride, dev_stride;
    const long long *u, *v, *nodes;
    const float *w;
    const unsigned char *active;
    const int *dev;
    int *comp;
    unsigned char *chosen, *ws;
};
template <bool LDS>
__global__ __launch_bounds__(T) void k_forest_batch(const ForestBatch b, int when) {
    const int s = blockIdx.x;
    const long s0 = b.start[s], e0 = s0 * b.stride;
    const int n = b.size[s], E = n * b.stride;
    u64 *best = reinterpret_cast<u64 *>(b.ws + b.ws_off[s]);
    int *slot = reinterpret_cast<int *>(best + n);
    int *su = slot + n, *sv = su + E, *gcomp = sv + E, *parent = gcomp + n, *rootof = parent + n;
    forest_body<LDS>(n, E, b.u + e0, b.v + e0, b.w ? b.w + e0 : nullptr, b.active ? b.active + e0 : nullptr, b.nodes + s0, n, b.comp + s0, slot, su,
                     sv, best, gcomp, parent, rootof, b.chosen ? b.chosen + e0 : nullptr, b.dev + (long)s * b.dev_stride, when);
}

// ---- two-component 1-D Gaussian mixture -----------------------------------------------------------------------------------------
constexpr int TG = 256;
constexpr int GMM_R = 32;   // values per thread the mixture fit keeps in registers (m <= GMM_R * TG = 8,192; more: re-read from memory)

template <int K>
__device__ inline void block_sum(double (&val)[K], double *lds) {   // -> every thread holds the sums (fixed order: reproducible)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = val[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if ((threadIdx.x & 63) == 0) lds[(threadIdx.x >> 6) * K + k] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
#pragma unroll
        for (int wv = 0; wv < TG / 64; ++wv) s += lds[wv * K + k];
        val[k] = s;
    }
}

__device__ inline double quantile_sorted(const float *xs, int m, double q) {   // numpy.quantile(method="linear") on sorted data
    const double pos = q * (double)(m - 1);
    int lo = (int)floor(pos);
    if (lo > m - 1) lo = m - 1;
    const int hi = lo + 1 < m ? lo + 1 : m - 1;
    const double g = pos - (double)lo, a = (double)xs[lo], b = (double)xs[hi];
    return g >= 0.5 ? b - (b - a) * (1.0 - g) : a + (b - a) * g;
}

// xs: m values sorted ascending; resp: 2 m doubles of scratch; out: mu0, mu1, var0, var1, pi0, pi1, iterations, log-likelihood.
struct Fit { double mu0, mu1, var0, var1; };   // (what every thread of the workgroup holds when the fit returns)

__device__ __forceinline__ Fit gmm2_fit(int m, const float *__restrict__ xs, double *resp, double *out, int iters, double tol, double reg,
                                        double *lds /* [(TG / 64) * 7] */) {
    const int t = threadIdx.x;
    if (m < 2 || xs[m - 1] == xs[0]) {
        double s[1] = {0.0};
        for (int i = t; i < m; i += TG) s[0] += (double)xs[i];
        block_sum<1>(s, lds);
        const double mean = m ? s[0] / (double)m : 0.0;
        if (t == 0) {
            out[0] = out[1] = mean;
            out[2] = out[3] = reg;
            out[4] = out[5] = 0.5;
            out[6] = 0.0;
            out[7] = 0.0;
        }
        return Fit{mean, mean, reg, reg};
    }
    const double q1 = quantile_sorted(xs, m, 0.25), q3 = quantile_sorted(xs, m, 0.75);
    double mu0 = q3 > q1 ? q1 : (double)xs[0], mu1 = q3 > q1 ? q3 : (double)xs[m - 1];
    double var0 = reg, var1 = reg, pi0 = 0.5, pi1 = 0.5, prev = -INFINITY, ll = 0.0;
    int it = 0;
    // One pass over the values and ONE block reduction per EM iteration (round 5; rounds 4-5 made three of each: the M-step's sums, the
    // variances around the new means, the E-step's log-likelihood -- and kept the responsibilities in a scratch array between them): the
    // E-step accumulates, with the responsibilities it has just computed, what the NEXT M-step needs -- sum r, sum r (x - c), sum r (x - c)^2
    // around the CURRENT means c -- so  mu' = c + sum r (x - c) / n  and  var' = sum r (x - c)^2 / n - (mu' - c)^2  (= sum r (x - mu')^2 / n:
    // the shift keeps the subtraction harmless, c and mu' differ by the step of one iteration).  Same iteration sequence and stopping rule.
    // Values: up to GMM_R per thread in registers (the pass's trees have ~2,400 edges), else re-read from xs (L2-resident).
    const bool inreg = m <= GMM_R * TG;
    float xr[GMM_R];
#pragma unroll
    for (int k = 0; k < GMM_R; ++k) xr[k] = (inreg && t + k * TG < m) ? xs[t + k * TG] : 0.f;
    auto for_each = [&](auto &&f) {   // thread t owns i = t, t + TG, ... in this order (fixed summation order)
        if (inreg) {
#pragma unroll
            for (int k = 0; k < GMM_R; ++k)
                if (t + k * TG < m) f((double)xr[k]);
        } else {
            for (int i = t; i < m; i += TG) f((double)xs[i]);
        }
    };
    for (int it2 = 0; it2 < 300; ++it2) {   // 2-means from the quartiles, run to its fixed point (sklearn's k-means start: max_iter 300)
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for_each([&](double x) {
            const int a = fabs(x - mu1) < fabs(x - mu0);   // (argmin: the first on a tie)
            s[a] += 1.0;
            s[2 + a] += x;
        });
        block_sum<4>(s, lds);
        if (s[0] == 0.0 || s[1] == 0.0) break;
        const double n0 = s[2] / s[0], n1 = s[3] / s[1];
        const bool fixed = n0 == mu0 && n1 == mu1;   // (every thread holds the same sums: a uniform branch)
        mu0 = n0;
        mu1 = n1;
        if (fixed) break;
    }
    double c0 = mu0, c1 = mu1;   // the shifts of the running sums
    double S[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for_each([&](double x) {     // hard assignment = the responsibilities the first M-step starts from
        const int a = fabs(x - mu1) < fabs(x - mu0);
        const double r0 = a ? 0.0 : 1.0, r1 = a ? 1.0 : 0.0, d0 = x - c0, d1 = x - c1;
        S[0] += r0; S[1] += r1; S[2] += r0 * d0; S[3] += r1 * d1; S[4] += r0 * d0 * d0; S[5] += r1 * d1 * d1;
    });
    block_sum<7>(S, lds);
    for (; it < iters; ++it) {
        const double n0 = S[0] + 1e-300, n1 = S[1] + 1e-300;
        pi0 = n0 / (double)m;
        pi1 = n1 / (double)m;
        mu0 = c0 + S[2] / n0;
        mu1 = c1 + S[3] / n1;
        var0 = S[4] / n0 - (mu0 - c0) * (mu0 - c0) + reg;
        var1 = S[5] / n1 - (mu1 - c1) * (mu1 - c1) + reg;
        const double k0 = -0.5 * log(2.0 * M_PI * var0) + log(pi0), k1 = -0.5 * log(2.0 * M_PI * var1) + log(pi1);
        double T[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for_each([&](double x) {
            const double d0 = x - mu0, d1 = x - mu1;
            const double p0 = k0 - 0.5 * d0 * d0 / var0, p1 = k1 - 0.5 * d1 * d1 / var1;
            // log-sum-exp and the two responsibilities from ONE exponential: with e = exp(-|p0 - p1|), lse = max + log(1 + e), the larger
            // responsibility is 1 / (1 + e) and the smaller e / (1 + e) (rounds 4-5 evaluated four exponentials and a logarithm per value:
            // the fit is issue-bound on one CU's double-precision pipes, 8 us per iteration at 2,400 values)
            const double mx = fmax(p0, p1), e = exp(-fabs(p0 - p1)), inv = 1.0 / (1.0 + e);
            const double lse = mx + log1p(e);
            const double rb = inv, rs = e * inv;
            const double r0 = p0 >= p1 ? rb : rs, r1 = p0 >= p1 ? rs : rb;
            T[0] += r0; T[1] += r1; T[2] += r0 * d0; T[3] += r1 * d1; T[4] += r0 * d0 * d0; T[5] += r1 * d1 * d1; T[6] += lse;
        });
        block_sum<7>(T, lds);
        ll = T[6] / (double)m;
#pragma unroll
        for (int k = 0; k < 6; ++k) S[k] = T[k];
        c0 = mu0; c1 = mu1;
        if (fabs(ll - prev) < tol) {
            ++it;
            break;
        }
        prev = ll;
    }
    {   // sklearn's loop is (E-step, M-step, test): the parameters it returns include the M-step of the LAST responsibilities
        const double n0 = S[0] + 1e-300, n1 = S[1] + 1e-300;
        pi0 = n0 / (double)m;
        pi1 = n1 / (double)m;
        mu0 = c0 + S[2] / n0;
        mu1 = c1 + S[3] / n1;
        var0 = S[4] / n0 - (mu0 - c0) * (mu0 - c0) + reg;
        var1 = S[5] / n1 - (mu1 - c1) * (mu1 - c1) + reg;
    }
    (void)resp;   // (scratch of the earlier forms: kept in the signature)
    if (t == 0) {
        out[0] = mu0;
        out[1] = mu1;
        out[2] = var0;
        out[3] = var1;
        out[4] = pi0;
        out[5] = pi1;
        out[6] = (double)it;
        out[7] = ll;
    }
    return Fit{mu0, mu1, var0, var1};
}

__global__ __launch_bounds__(TG) void k_gmm2(int m, const float *__restrict__ xs, double *resp, double *out, int iters, double tol,
                                             double reg, const int *__restrict__ m_dev = nullptr) {
    __shared__ double lds[(TG / 64) * 7];
    if (m_dev) m = min(m, *m_dev);   // (m by value = the capacity of xs / resp)
    gmm2_fit(m, xs, resp, out, iters, tol, reg, lds);
}

// Every scene of a batch (grid = scenes): the fit of the scene's first tdev[2 s + 1] sorted weights, then upstream's cut of the tree
// (pointpdf_v1m1_base.py:346-358): the component with the larger mean, "std" = its covariance as upstream, weak[e] = tw[e] < mean - 2 * "std"
// for the scene's tdev[2 s + 1] tree entries in entry order.
__global__ __launch_bounds__(TG) void k_gmm2_weak(const int *__restrict__ starts, const int *__restrict__ sizes, const int *__restrict__ tdev,
                                                  const float *__restrict__ xs, const float *__restrict__ tw, double *resp, double *fit,
                                                  unsigned char *__restrict__ weak, int iters, double tol, double reg) {
    __shared__ double lds[(TG / 64) * 7];
    const int s = blockIdx.x;
    const long s0 = starts[s];
    const int n = sizes[s], m = min(tdev[2 * s + 1], n);
    const Fit f = gmm2_fit(m, xs + s0, resp + 2 * s0, fit + 8 * s, iters, tol, reg, lds);
    const double lower = f.mu1 > f.mu0 ? f.mu1 - 2.0 * f.var1 : f.mu0 - 2.0 * f.var0;   // (np.argmax(means): the first on a tie)
    for (int i = threadIdx.x; i < m; i += TG) weak[s0 + i] = (double)tw[s0 + i] < lower ? 1 : 0;   // (the second labelling reads tdev[2 s + 1] entries)
}

}   // namespace gp

extern "C" long pdf_graph_forest_workspace_bytes(long n, long E, long n_nodes) {
    if (n < 0 || E < 0 || n_nodes < 0) return 0;
    return n_nodes * (long)sizeof(gp::u64) + (n + 2 * E + 3 * n_nodes) * (long)sizeof(int);
}

// Minimum spanning forest / connected components of one scene's region graph: see k_forest.  comp: n ints (out, written at the listed
// nodes only); chosen: E bytes or NULL; w, active: may be NULL; workspace: pdf_graph_forest_workspace_bytes(n, E, n_nodes), 8-byte aligned.
extern "C" int pdf_graph_forest(long n, int E, const long long *u, const long long *v, const float *w, const unsigned char *active,
                                const long long *nodes, int n_nodes, int *comp, unsigned char *chosen, void *workspace,
                                long workspace_bytes, void *stream) {
    if (n < 0 || E < 0 || n_nodes < 0 || n > 0x7fffffffL) return PDF_ERR_BAD_ARG;
    if (n_nodes == 0) return PDF_OK;
    if (!nodes || !comp || !workspace || (E > 0 && (!u || !v))) return PDF_ERR_BAD_ARG;
    if (workspace_bytes < pdf_graph_forest_workspace_bytes(n, E, n_nodes) || (reinterpret_cast<uintptr_t>(workspace) & 7)) return PDF_ERR_BAD_ARG;
    gp::u64 *best = static_cast<gp::u64 *>(workspace);
    int *slot = reinterpret_cast<int *>(best + n_nodes);
    int *su = slot + n, *sv = su + E, *gcomp = sv + E, *parent = gcomp + n_nodes, *rootof = parent + n_nodes;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // The LDS form needs up to LDS_NODES * 12 bytes of dynamic LDS (> the 64 KB default limit): raised ONCE per device to the maximum any
    // call can ask for (a per-call value would race between host threads: one lowers the limit another is about to launch with), result
    // checked -- where the runtime refuses, every call takes the global-memory form.
    static PdfLdsLimit lds_site;
    const bool lds_form = pdf_lds_limit_raised(lds_site, reinterpret_cast<const void *>(&gp::k_forest<true>),
                                               (int)(gp::LDS_NODES * (sizeof(gp::u64) + sizeof(int))));
    if (n_nodes <= gp::LDS_NODES && (lds_form || (size_t)n_nodes * (sizeof(gp::u64) + sizeof(int)) <= 64 * 1024)) {
        const size_t lds = (size_t)n_nodes * (sizeof(gp::u64) + sizeof(int));
        gp::k_forest<true><<<1, gp::T, lds, s>>>(n, E, u, v, w, active, nodes, n_nodes, comp, slot, su, sv, best, gcomp, parent, rootof, chosen);
    } else {
        gp::k_forest<false><<<1, gp::T, 0, s>>>(n, E, u, v, w, active, nodes, n_nodes, comp, slot, su, sv, best, gcomp, parent, rootof, chosen);
    }
    return pdf_launch_status();
}

#ifdef GP_PROFILE
extern "C" int pdf_graph_forest_profile(long long *host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gp::gp_prof), sizeof(long long) * 64 * 8) == hipSuccess ? PDF_OK : PDF_ERR_BAD_ARG;
}
#endif

// Two-component 1-D Gaussian mixture of m sorted float values by EM in double (sklearn.mixture.GaussianMixture(n_components=2,
// reg_covar=reg, tol=tol, max_iter=iters) with a deterministic start: quartiles -> 2-means).  resp: 2 m doubles of scratch;
// out (8 doubles): means, variances, weights of the two components, the iterations run, the final mean log-likelihood.
// The same graph with its sizes in DEVICE memory (dev = [n_nodes, E]; the by-value E / n_nodes are capacities): nothing is read back, so the
// pseudo-label pass stays free of host syncs (and capturable).  Both forms of the kernel are launched; the one the actual size does not
// call for returns at once.  workspace: pdf_graph_forest_workspace_bytes(n, E capacity, n_nodes capacity).
extern "C" int pdf_graph_forest_dev(long n, int E, const long long *u, const long long *v, const float *w, const unsigned char *active,
                                    const long long *nodes, int n_nodes, const int *dev, int *comp, unsigned char *chosen, void *workspace,
                                    long workspace_bytes, void *stream) {
    if (n < 0 || E < 0 || n_nodes < 0 || n > 0x7fffffffL || !dev) return PDF_ERR_BAD_ARG;
    if (n_nodes == 0) return PDF_OK;
    if (!nodes || !comp || !workspace || (E > 0 && (!u || !v))) return PDF_ERR_BAD_ARG;
    if (workspace_bytes < pdf_graph_forest_workspace_bytes(n, E, n_nodes) || (reinterpret_cast<uintptr_t>(workspace) & 7)) return PDF_ERR_BAD_ARG;
    gp::u64 *best = static_cast<gp::u64 *>(workspace);
    int *slot = reinterpret_cast<int *>(best + n_nodes);
    int *su = slot + n, *sv = su + E, *gcomp = sv + E, *parent = gcomp + n_nodes, *rootof = parent + n_nodes;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static PdfLdsLimit lds_site;
    const bool lds_form = pdf_lds_limit_raised(lds_site, reinterpret_cast<const void *>(&gp::k_forest<true>),
                                               (int)(gp::LDS_NODES * (sizeof(gp::u64) + sizeof(int))));
    if (lds_form) {
        const size_t lds = (size_t)std::min(n_nodes, gp::LDS_NODES) * (sizeof(gp::u64) + sizeof(int));
        gp::k_forest<true><<<1, gp::T, lds, s>>>(n, E, u, v, w, active, nodes, n_nodes, comp, slot, su, sv, best, gcomp, parent, rootof, chosen, dev, 1);
        if (n_nodes > gp::LDS_NODES)
            gp::k_forest<false><<<1, gp::T, 0, s>>>(n, E, u, v, w, active, nodes, n_nodes, comp, slot, su, sv, best, gcomp, parent, rootof, chosen, dev, 2);
    } else {
        gp::k_forest<false><<<1, gp::T, 0, s>>>(n, E, u, v, w, active, nodes, n_nodes, comp, slot, su, sv, best, gcomp, parent, rootof, chosen, dev, 0);
    }
    return pdf_launch_status();
}

extern "C" int pdf_gmm2_1d_dev(int m_cap, const float *sorted_x, const int *m_dev, double *resp, double *out, int iters, double tol, double reg,
                               void *stream) {
    if (m_cap < 0 || !out || !m_dev || (m_cap > 0 && (!sorted_x || !resp)) || iters < 1) return PDF_ERR_BAD_ARG;
    gp::k_gmm2<<<1, gp::TG, 0, static_cast<hipStream_t>(stream)>>>(m_cap, sorted_x, resp, out, iters, tol, reg, m_dev);
    return pdf_launch_status();
}

extern "C" int pdf_gmm2_1d(int m, const float *sorted_x, double *resp, double *out, int iters, double tol, double reg, void *stream) {
    if (m < 0 || !out || (m > 0 && (!sorted_x || !resp)) || iters < 1) return PDF_ERR_BAD_ARG;
    gp::k_gmm2<<<1, gp::TG, 0, static_cast<hipStream_t>(stream)>>>(m, sorted_x, resp, out, iters, tol, reg);
    return pdf_launch_status();
}

// pdf_gmm2_1d_dev for every scene of a batch + the cut of the spanning tree it decides (see k_gmm2_weak).  sorted_x, tw (N) floats; resp 2 N
// doubles of scratch; fit (scenes, 8) doubles out; weak (N) bytes out.
extern "C" int pdf_gmm2_weak_dev(int scenes, const int *starts, const int *sizes, const int *tdev, const float *sorted_x, const float *tw,
                                 double *resp, double *fit, unsigned char *weak, int iters, double tol, double reg, void *stream) {
    if (scenes < 0 || iters < 1) return PDF_ERR_BAD_ARG;
    if (scenes == 0) return PDF_OK;
    if (!starts || !sizes || !tdev || !sorted_x || !tw || !resp || !fit || !weak) return PDF_ERR_BAD_ARG;
    gp::k_gmm2_weak<<<scenes, gp::TG, 0, static_cast<hipStream_t>(stream)>>>(starts, sizes, tdev, sorted_x, tw, resp, fit, weak, iters, tol, reg);
    return pdf_launch_status();
}

// pdf_graph_forest_dev for the scenes of a batch in one launch per form (see k_forest_batch).  starts / sizes: HOST arrays (scenes ints);
// entry arrays u, v, w, active, chosen (N * stride) and nodes, comp (N) are the batch's, scene s at starts[s] (* stride); dev: the scenes'
// [nodes, entries] pairs dev_stride ints apart; workspace: sum over the scenes of pdf_graph_forest_workspace_bytes(size, size * stride, size)
// rounded up to 8 bytes each.
extern "C" int pdf_graph_forest_batch_dev(int scenes, const int *starts, const int *sizes, int stride, const long long *u, const long long *v,
                                          const float *w, const unsigned char *active, const long long *nodes, const int *dev, int dev_stride,
                                          int *comp, unsigned char *chosen, void *workspace, long workspace_bytes, void *stream) {
    if (scenes < 0 || stride < 1 || dev_stride < 2) return PDF_ERR_BAD_ARG;
    if (scenes == 0) return PDF_OK;
    if (!starts || !sizes || !u || !v || !nodes || !dev || !comp || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 7)) return PDF_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    static PdfLdsLimit lds_site;
    const bool lds_form = pdf_lds_limit_raised(lds_site, reinterpret_cast<const void *>(&gp::k_forest_batch<true>),
                                               (int)(gp::LDS_NODES * (sizeof(gp::u64) + sizeof(int))));
    long off = 0;
    for (int at = 0; at < scenes; at += gp::FOREST_BATCH) {
        gp::ForestBatch b;
        const int nb = std::min(scenes - at, gp::FOREST_BATCH);
        int nmax = 0;
        for (int i = 0; i < nb; ++i) {
            if (sizes[at + i] < 0 || starts[at + i] < 0) return PDF_ERR_BAD_ARG;
            b.start[i] = starts[at + i]; b.size[i] = sizes[at + i]; b.ws_off[i] = off;
            off += (pdf_graph_forest_workspace_bytes(sizes[at + i], (long)sizes[at + i] * stride, sizes[at + i]) + 7) & ~7L;
            nmax = std::max(nmax, sizes[at + i]);
        }
        if (off > workspace_bytes) return PDF_ERR_BAD_ARG;
        b.stride = stride; b.dev_stride = dev_stride; b.u = u; b.v = v; b.nodes = nodes; b.w = w; b.active = active;
        b.dev = dev + (long)at * dev_stride; b.comp = comp; b.chosen = chosen; b.ws = static_cast<unsigned char *>(workspace);
        if (lds_form) {
            const size_t lds = (size_t)std::min(nmax, gp::LDS_NODES) * (sizeof(gp::u64) + sizeof(int));
            gp::k_forest_batch<true><<<nb, gp::T, lds, s>>>(b, 1);
            if (nmax > gp::LDS_NODES) gp::k_forest_batch<false><<<nb, gp::T, 0, s>>>(b, 2);
        } else {
            gp::k_forest_batch<false><<<nb, gp::T, 0, s>>>(b, 0);
        }
    }
    return pdf_launch_status();
}
