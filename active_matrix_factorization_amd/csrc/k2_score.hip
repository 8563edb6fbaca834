// k2_score.hip - candidate-grid variance scoring under the matrix-normal posterior.
//
// Replaces the per-candidate Python loop  pred_variance -> approx_pred_mean_var ->
// exp_dotprod_sq  (mn_active_pmf.py:300-330,501-511; matrix_normal_exps_cy.pyx:126-154).
// The reference sums d(d+1)/2 Isserlis terms per candidate and subtracts mean^2.  With
// a = M_i, b = M_{n+j}, s_ii, s_jj, s_ij the entries of Sigma = cov_useritems and
// O = cov_latents (the reference only ever reads O[k,l] with k <= l, so O is taken as its
// upper triangle mirrored), the same quantity is
//     Var = (s_ii s_jj + s_ij^2) tr(O^2) + s_jj a'Oa + s_ii b'Ob + 2 s_ij (Oa)'b
//     E   = a'b + s_ij tr(O)
// i.e. per candidate ONE streamed operand (s_ij) and a rank-d dot product against
// per-row / per-column vectors.  The sweep is HBM-bound: 4 B read + 4 B written per
// candidate in f32 (8 + 8 in f64).
//
// Two float32/float64 kernels share the packs written by k_mn_prep:
//  * k_mn_grid (rank <= 12, and every float64 sweep): a block is 4 waves that share a chunk of
//    64*VEC columns; each lane keeps its VEC columns' b_j (D x VEC registers), q_j, s_jj for the
//    whole block.  A wave walks its own 16 rows: the wave-uniform row operands (Oa_i, a_i,
//    a'Oa, s_ii) are fetched once with coalesced loads and broadcast from registers
//    (v_readlane), the s_ij row segment comes through one 16-byte nontemporal load per lane
//    and the scores leave through one 16-byte nontemporal store per lane.  No LDS.
//  * k_mn_grid_mfma (float32, rank >= 13, and reduction-only sweeps at any rank): the rank-d
//    term on the matrix cores; see the comment above that kernel.
#include "common.h"

namespace {

constexpr int ROWS_PER_WAVE = 16;
constexpr int WAVES = 4;
constexpr int ROWS_PER_BLOCK = ROWS_PER_WAVE * WAVES;
#ifndef APMF_K2_UNR
#define APMF_K2_UNR 4
#endif
constexpr int UNR = APMF_K2_UNR;

template <typename T, int N> struct Vec;
template <> struct Vec<float, 4> { using type = float4; };
template <> struct Vec<float, 2> { using type = float2; };
template <> struct Vec<float, 1> { using type = float; };
template <> struct Vec<double, 2> { using type = double2; };
template <> struct Vec<double, 1> { using type = double; };

template <typename T, int V>
__device__ __forceinline__ void vload(T (&dst)[V], const T *p)
{
    using VT = typename Vec<T, V>::type;
    VT v = *reinterpret_cast<const VT *>(p);
    const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
    for (int i = 0; i < V; i++) dst[i] = e[i];
}

template <typename T, int V>
__device__ __forceinline__ void vstore(T *p, const T (&src)[V])
{
    using VT = typename Vec<T, V>::type;
    VT v;
    T *e = reinterpret_cast<T *>(&v);
#pragma unroll
    for (int i = 0; i < V; i++) e[i] = src[i];
    *reinterpret_cast<VT *>(p) = v;
}

struct GridDims {
    int B, n, m, d, ld;
    int rs;            // rowpack row stride (elements)
    int nblk_x, nblk_y;
    int want;          // +1 max, -1 min, 0 none
    int want_sum;
    int csteps;        // MFMA sweep: tiles of 32 rows per block
};

// Operands are separate `const __restrict__` kernel parameters (not members of a struct)
// so that the wave-uniform row operands can be fetched with scalar loads.
template <typename T>
struct GridArgs {
    GridDims g;
    const T *cross;    // (B, n, ld)
    const T *rowpack;  // (B, n, rs): [Oa (D) | a (D) | a'Oa | s_ii]
    const T *colpack;  // (B, D+2, ld): b_k rows, then b'Ob, then s_jj
    const T *state;    // (B, 2): tr(O^2), tr(O)
    const uint16_t *maskT;  // (ceil(n/16), ld) bit r of word [g][j] = mask[16 g + r][j], or NULL
    T *var;            // (B, n, ld) or NULL
    T *pmean;          // (B, n, ld) or NULL
    double *part_sum;     // (B, nblk)
    T *part_val;          // (B, nblk)
    long long *part_idx;  // (B, nblk)
};

// ---- per-state operand packs ------------------------------------------------------
// One thread per row of M (users -> rowpack, items -> colpack).  The block's rows of M are
// staged in LDS (coalesced fetch, conflict-free per-row reads) and then held in registers
// (DT = compile-time latent-dimension bound, 0 = generic loop over LDS); Omega entries are
// wave-uniform operands (scalar loads); padding columns of colpack are zeroed.
template <typename T, int DT>
__global__ __launch_bounds__(256) void k_mn_prep(int B, int n, int m, int d, int D, int ld, int rs,
                                                 int tiled, int stage, const T *__restrict__ mean,
                                                 const T *__restrict__ diag,
                                                 const T *__restrict__ omega,
                                                 T *__restrict__ rowpack, T *__restrict__ colpack,
                                                 T *__restrict__ state)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sA = reinterpret_cast<T *>(smem_raw);   // blockDim.x x st
    const int st = d | 1;                      // odd stride: conflict-free row reads
    const int R = blockDim.x;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * R;             // first row (of the n + ld row space) of this block
    const int t = t0 + threadIdx.x;
    const T *__restrict__ O = omega + (size_t)b * d * d;
    {
        const int nreal = max(0, min(R, n + m - t0));
        const T *src = mean + ((size_t)b * (n + m) + t0) * d;
        const int total = nreal * d;
        if (DT > 0 && total > 0) {      // (a block of padding columns only has nothing to stage: src is past the array)
            // all of the thread's elements requested before the first is stored (a plain loop is one
            // round trip per element: d of them in a row)
            T v[DT > 0 ? DT : 1];
#pragma unroll
            for (int it = 0; it < DT; it++) {
                const int e = (int)threadIdx.x + it * R;
                v[it] = src[e < total ? e : 0];
            }
#pragma unroll
            for (int it = 0; it < DT; it++) {
                const int e = (int)threadIdx.x + it * R;
                if (it < d && e < total) {
                    const int r = e / d, k = e - r * d;
                    sA[r * st + k] = v[it];
                }
            }
        } else if (DT == 0) {
            for (int e = threadIdx.x; e < total; e += R) {
                const int r = e / d, k = e - r * d;
                sA[r * st + k] = src[e];
            }
        }
    }
    // block 0 of a state also reduces Omega to its two scalars: Omega staged in LDS first (read from
    // global memory inside the loop, the d^2 dependent round trips made this block the kernel's tail)
    T *sO = sA + (size_t)R * st;
    if (blockIdx.x == 0)
        for (int e = threadIdx.x; e < d * d; e += R) sO[e] = O[e];
    __syncthreads();
    if (t == 0) {
        T t2 = 0, tr = 0;
        for (int k = 0; k < d; k++) {
            tr += sO[k * d + k];
            t2 += sO[k * d + k] * sO[k * d + k];
            for (int l = k + 1; l < d; l++) t2 += 2 * sO[k * d + l] * sO[k * d + l];
        }
        state[2 * b] = t2;
        state[2 * b + 1] = tr;
    }
    // stage (row-major user pack only): a thread's rs values go to LDS first and the block then writes
    // its rows as one contiguous run - written directly, every store instruction of a wave scatters
    // over 64 rows rs * sizeof(T) bytes apart
    T *sP = sO + (size_t)d * d;
    do {
    if (t >= n + ld) break;
    const bool is_user = t < n;
    const int j = t - n;  // column when !is_user (may be padding >= m)
    T *cp = colpack + (size_t)b * (D + 2) * ld;
    if (!is_user && j >= m) {
        for (int k = 0; k < D + 2; k++) cp[(size_t)k * ld + j] = 0;
        break;
    }
    const T *a = sA + threadIdx.x * st;
    const T sd = diag[(size_t)b * (n + m) + t];
    T quad = 0;
    // user pack: row-major [Oa | a | a'Oa | s_ii] with stride rs, or (tiled, for the matrix-core
    // sweep) tile-major transposed: tile t/32 holds (2D+2) x 32 floats [Oa | a'Oa | s_ii | a]
    T *rp = tiled ? rowpack + (((size_t)b * ((n + 31) >> 5) + (t >> 5)) * (2 * D + 2)) * 32 + (t & 31)
                  : (stage ? sP + (size_t)threadIdx.x * (rs + 1) : rowpack + ((size_t)b * n + t) * rs);
    const size_t rstr = tiled ? 32 : 1;
    const int pos_a = tiled ? D + 2 : D, pos_q = tiled ? D : 2 * D;
    if (DT > 0) {
        // the row in registers, loops fully unrolled; the reference only reads Omega[k, l]
        // with k <= l, so the upper triangle is mirrored
        T av[DT > 0 ? DT : 1];
#pragma unroll
        for (int k = 0; k < DT; k++) av[k] = k < d ? a[k] : (T)0;
#pragma unroll
        for (int k = 0; k < DT; k++) {
            T w = 0;
            if (k < d) {
#pragma unroll
                for (int l = 0; l < DT; l++)
                    if (l < d) w += O[l < k ? l * d + k : k * d + l] * av[l];
                quad += av[k] * w;
            }
            if (k < D) {
                if (is_user) { rp[k * rstr] = w; rp[(pos_a + k) * rstr] = av[k]; }
                else cp[(size_t)k * ld + j] = av[k];
            }
        }
        for (int k = DT; k < D; k++) {
            if (is_user) { rp[k * rstr] = 0; rp[(pos_a + k) * rstr] = 0; }
            else cp[(size_t)k * ld + j] = 0;
        }
    } else {
        for (int k = 0; k < D; k++) {
            T w = 0, ak = 0;
            if (k < d) {
                ak = a[k];
                for (int l = 0; l < k; l++) w += O[l * d + k] * a[l];
                for (int l = k; l < d; l++) w += O[k * d + l] * a[l];
                quad += ak * w;
            }
            if (is_user) {
                rp[k * rstr] = w;
                rp[(pos_a + k) * rstr] = ak;
            } else {
                cp[(size_t)k * ld + j] = ak;
            }
        }
    }
    if (is_user) {
        rp[pos_q * rstr] = quad;
        rp[(pos_q + 1) * rstr] = sd;
    } else {
        cp[(size_t)D * ld + j] = quad;
        cp[(size_t)(D + 1) * ld + j] = sd;
    }
    } while (0);
    if (stage && !tiled) {           // block-uniform
        __syncthreads();
        const int nuser = max(0, min(R, n - t0));
        T *dst = rowpack + ((size_t)b * n + t0) * rs;
        for (int e = threadIdx.x; e < nuser * rs; e += R) dst[e] = sP[(size_t)(e / rs) * (rs + 1) + e % rs];
    }
}

// (n, ld) uint8 mask -> (ceil(n/16), ld) uint16 words, bit r = row 16 g + r: a lane fetches
// the candidate bits of its columns for all 16 rows of its wave with ONE small load.
// A thread packs 4 adjacent columns (one 4-byte load per row, one 8-byte store).
__global__ __launch_bounds__(256) void k_mask_pack(int n, int ld, const uint8_t *__restrict__ mask,
                                                   uint16_t *__restrict__ maskT)
{
    const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int g = blockIdx.y;
    if (j >= ld) return;
    unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    const int rows = min(16, n - g * 16);
    const uint8_t *src = mask + (size_t)g * 16 * ld + j;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        if (r < rows) {
            const unsigned v = *reinterpret_cast<const unsigned *>(src + (size_t)r * ld);
            w0 |= ((v & 0xFFu) ? 1u : 0u) << r;
            w1 |= ((v & 0xFF00u) ? 1u : 0u) << r;
            w2 |= ((v & 0xFF0000u) ? 1u : 0u) << r;
            w3 |= ((v & 0xFF000000u) ? 1u : 0u) << r;
        }
    }
    uint2 out;
    out.x = w0 | (w1 << 16);
    out.y = w2 | (w3 << 16);
    *reinterpret_cast<uint2 *>(maskT + (size_t)g * ld + j) = out;
}

// ---- the sweep ----------------------------------------------------------------------
#ifndef APMF_K2_PREFETCH
#define APMF_K2_PREFETCH 1
#endif
#ifndef APMF_K2_NT
#define APMF_K2_NT 1   // streamed-once operands use nontemporal loads/stores
#endif

template <typename T, int N> struct NVec { typedef T type __attribute__((ext_vector_type(N))); };
template <typename T> struct NVec<T, 1> { typedef T type; };

template <typename T, int V>
__device__ __forceinline__ void vload_stream(T (&dst)[V], const T *p)
{
    using VT = typename NVec<T, V>::type;
#if APMF_K2_NT
    VT v = __builtin_nontemporal_load(reinterpret_cast<const VT *>(p));
#else
    VT v = *reinterpret_cast<const VT *>(p);
#endif
    const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
    for (int i = 0; i < V; i++) dst[i] = e[i];
}

template <typename T, int V>
__device__ __forceinline__ void vstore_stream(T *p, const T (&src)[V])
{
    using VT = typename NVec<T, V>::type;
    VT v;
    T *e = reinterpret_cast<T *>(&v);
#pragma unroll
    for (int i = 0; i < V; i++) e[i] = src[i];
#if APMF_K2_NT
    __builtin_nontemporal_store(v, reinterpret_cast<VT *>(p));
#else
    *reinterpret_cast<VT *>(p) = v;
#endif
}

// Element `lane_idx` of a wave-resident register as a wave-uniform value (v_readlane): the
// per-row operands of a wave's 16 rows are fetched once with coalesced vector loads and then
// broadcast from registers, instead of 16 x dependent scalar-cache round trips.
__device__ __forceinline__ float bcast_lane(float v, int lane_idx)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_idx));
}
__device__ __forceinline__ double bcast_lane(double v, int lane_idx)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane_idx);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane_idx);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <typename T, int D, int VEC, bool RED, bool MEAN>
__global__ __launch_bounds__(256) void k_mn_grid(const GridDims a, const T *__restrict__ p_cross,
                                                 const T *__restrict__ p_rowpack,
                                                 const T *__restrict__ p_colpack,
                                                 const T *__restrict__ p_state,
                                                 const uint16_t *__restrict__ p_maskT,
                                                 T *__restrict__ p_var, T *__restrict__ p_pmean,
                                                 double *__restrict__ part_sum,
                                                 T *__restrict__ part_val,
                                                 long long *__restrict__ part_idx)
{
    static_assert(ROWS_PER_WAVE == 16, "mask words hold 16 rows");
    const int lane = threadIdx.x & 63;
    const int wave = wave_id_uniform();
    // XCD-aware decode of the 1-D grid: workgroups are dealt round-robin over the 8 XCDs, so
    // all tiles of one posterior state go to the same XCD and its packs stay in ONE L2.
    // (Placement is a speed matter only: any mapping covers every (state, tile) exactly once.)
    // States beyond the last full group of 8 (all of them when B < 8) have their tiles dealt
    // over all XCDs instead, so that a single-state sweep still fills the chip.
    const int per_state = a.nblk_x * a.nblk_y;
    const int L = blockIdx.x;
    const int full = (a.B >> 3) * per_state * 8;
    int b, tile;
    if (L < full) {
        const int xcd = L & 7, q = L >> 3;
        b = (q / per_state) * 8 + xcd;
        tile = q % per_state;
    } else {
        const int q = L - full;
        b = (a.B & ~7) + q / per_state;
        tile = q % per_state;
    }
    const int bx = tile % a.nblk_x, by = tile / a.nblk_x;
    const int j0 = (bx * 64 + lane) * VEC;
    const bool colok = j0 < a.ld;
    const int ld = a.ld, n = a.n, m = a.m;
    const int rbeg = by * ROWS_PER_BLOCK + wave * ROWS_PER_WAVE;
    const int rend = min(rbeg + ROWS_PER_WAVE, n);

    T bt[D][VEC], qq[VEC], sjj[VEC];
    {
        const T *cp = p_colpack + (size_t)b * (D + 2) * ld + j0;
#pragma unroll
        for (int k = 0; k < D; k++) {
            if (colok) vload<T, VEC>(bt[k], cp + (size_t)k * ld);
            else
#pragma unroll
                for (int v = 0; v < VEC; v++) bt[k][v] = 0;
        }
        if (colok) {
            vload<T, VEC>(qq, cp + (size_t)D * ld);
            vload<T, VEC>(sjj, cp + (size_t)(D + 1) * ld);
        } else {
#pragma unroll
            for (int v = 0; v < VEC; v++) qq[v] = sjj[v] = 0;
        }
    }
    const T t2 = p_state[2 * b], trO = p_state[2 * b + 1];
    // RED / MEAN are compile-time so that the plain "write the scores" sweep does not carry
    // the registers of the reductions and of the second output
    constexpr bool do_mean = MEAN;
    const bool do_var = p_var != nullptr;
    constexpr bool do_red = RED;

    // the wave's 16 rows of rowpack (contiguous: RS elements per row), element e in
    // register e / 64, lane e % 64
    constexpr int RS = (2 * D + 2 + 3) / 4 * 4;
    constexpr int NRP = (ROWS_PER_WAVE * RS + 63) / 64;
    T rpk[NRP];
    {
        const T *rp0 = p_rowpack + ((size_t)b * n + rbeg) * RS;
        const int nel = (rend - rbeg) * RS;
#pragma unroll
        for (int c = 0; c < NRP; c++) rpk[c] = (c * 64 + lane < nel) ? rp0[c * 64 + lane] : (T)0;
    }
#define ROWOP(r, k) bcast_lane(rpk[((r) * RS + (k)) / 64], ((r) * RS + (k)) % 64)

    // candidate bits of this lane's columns for the wave's 16 rows
    unsigned mbits[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) mbits[v] = (colok && j0 + v < m) ? 0xFFFFu : 0u;
    if (p_maskT && colok && do_red && rbeg < n) {
        const uint16_t *mp = p_maskT + (size_t)(rbeg >> 4) * ld + j0;
        uint16_t w[VEC];
        __builtin_memcpy(w, __builtin_assume_aligned(mp, 2 * VEC), sizeof(w));
#pragma unroll
        for (int v = 0; v < VEC; v++) mbits[v] &= w[v];
    }

    double sum = 0.;
    T bestv = 0;
    int bestrow = -1, bestcol = 0;

    T s_cur[UNR][VEC];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
        if (rbeg + u < rend && colok) vload_stream<T, VEC>(s_cur[u], p_cross + ((size_t)b * n + rbeg + u) * ld + j0);
        else
#pragma unroll
            for (int v = 0; v < VEC; v++) s_cur[u][v] = 0;
    }

#pragma unroll
    for (int g = 0; g < ROWS_PER_WAVE / UNR; g++) {
        const int i0 = rbeg + g * UNR;
        if (i0 >= rend) break;  // wave-uniform
#if APMF_K2_PREFETCH
        // prefetch the next UNR rows while this group is computed
        T s_nxt[UNR][VEC];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int i = i0 + UNR + u;
            if (i < rend && colok) vload_stream<T, VEC>(s_nxt[u], p_cross + ((size_t)b * n + i) * ld + j0);
            else
#pragma unroll
                for (int v = 0; v < VEC; v++) s_nxt[u][v] = 0;
        }
#else
        if (g > 0) {
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int i = i0 + u;
                if (i < rend && colok) vload_stream<T, VEC>(s_cur[u], p_cross + ((size_t)b * n + i) * ld + j0);
            }
        }
#endif
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int r = g * UNR + u;   // compile-time row slot of this wave
            const int i = i0 + u;
            if (i >= rend) break;  // wave-uniform
            const T p = ROWOP(r, 2 * D), sii = ROWOP(r, 2 * D + 1);
            T dot[VEC], var[VEC];
#pragma unroll
            for (int v = 0; v < VEC; v++) dot[v] = 0;
#pragma unroll
            for (int k = 0; k < D; k++) {
                const T wk = ROWOP(r, k);
#pragma unroll
                for (int v = 0; v < VEC; v++) dot[v] += wk * bt[k][v];
            }
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                const T sv = s_cur[u][v];
                var[v] = (sii * sjj[v] + sv * sv) * t2 + sjj[v] * p + sii * qq[v] + 2 * sv * dot[v];
            }
            if (colok) {
                if (do_var) vstore_stream<T, VEC>(p_var + ((size_t)b * n + i) * ld + j0, var);
                if (do_mean) {
                    T mn[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; v++) mn[v] = 0;
#pragma unroll
                    for (int k = 0; k < D; k++) {
                        const T ak = ROWOP(r, D + k);
#pragma unroll
                        for (int v = 0; v < VEC; v++) mn[v] += ak * bt[k][v];
                    }
#pragma unroll
                    for (int v = 0; v < VEC; v++) mn[v] += s_cur[u][v] * trO;
                    vstore_stream<T, VEC>(p_pmean + ((size_t)b * n + i) * ld + j0, mn);
                }
            }
            if (do_red) {
                T rowsum = 0;
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    const bool cand = (mbits[v] >> r) & 1u;
                    rowsum += cand ? var[v] : (T)0;
                    if (a.want) {
                        // rows then columns ascend within a lane: a strict compare keeps the
                        // first occurrence
                        const bool better = cand && (bestrow < 0 || (a.want > 0 ? var[v] > bestv : var[v] < bestv));
                        bestv = better ? var[v] : bestv;
                        bestrow = better ? i : bestrow;
                        bestcol = better ? v : bestcol;
                    }
                }
                sum += (double)rowsum;
            }
        }
#if APMF_K2_PREFETCH
#pragma unroll
        for (int u = 0; u < UNR; u++)
#pragma unroll
            for (int v = 0; v < VEC; v++) s_cur[u][v] = s_nxt[u][v];
#endif
    }
#undef ROWOP

    if (do_red) {
        __shared__ double sh_sum[WAVES];
        __shared__ T sh_val[WAVES];
        __shared__ long long sh_idx[WAVES];
        long long besti = bestrow < 0 ? -1 : (long long)bestrow * m + (j0 + bestcol);
        sum = wave_sum(sum);
        if (a.want) wave_argext(bestv, besti, a.want);
        if (lane == 0) { sh_sum[wave] = sum; sh_val[wave] = bestv; sh_idx[wave] = besti; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double ts = sh_sum[0];
            T bv = sh_val[0];
            long long bi = sh_idx[0];
            for (int w = 1; w < WAVES; w++) {
                ts += sh_sum[w];
                if (a.want) argext_merge(bv, bi, sh_val[w], sh_idx[w], a.want);
            }
            const size_t slot = (size_t)b * per_state + tile;
            part_sum[slot] = ts;
            part_val[slot] = bv;
            part_idx[slot] = bi;
        }
    }
}

// ---- the sweep for d > 16, float32: the rank-d term on the matrix cores ----------------
// At d = 64 a candidate costs 2d = 128 flops against 8 bytes: the sweep is no longer a pure
// stream, the bilinear term G = (A Omega) B' IS a GEMM (n x d by d x m), and the f32 MFMA
// (v_mfma_f32_32x32x2_f32: exact f32, an fmaf chain) does it at the full f32 rate while the
// VALU is left to the epilogue.
// A block is 4 waves x 32 columns = 128 adjacent item columns (512 contiguous bytes of every
// grid row it touches).  Each wave keeps ITS 32 items' b_j in registers as the MFMA B operand
// for the whole walk (plus q_j, s_jj), and the block walks `csteps` tiles of 32 user rows; the
// user-side operands of a tile (Omega a, a'Oa, s_ii, and a for the mean output) are staged
// through LDS, double-buffered, shared by the 4 waves.  They come from `rowT`, the tile-major
// transposed pack written by k_mn_prep (tile t: (2D+2) x 32 floats, row k of it contiguous),
// so staging is one contiguous copy and every LDS read of the MFMA loop is conflict-free.
// k is permuted so that lane half h = lane >> 5 owns k in [h D/2, (h+1) D/2).
// Accumulator layout (32x32): col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5):
// a lane owns ONE item column for the whole walk and visits its rows in ascending order, so
// the arg-extremum needs only a strict compare to keep the lowest flat index on ties.
constexpr int MF_BCOLS = 128;   // columns per block (4 waves x 32)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory
// (s_waitcnt vmcnt(0)): inside the tile loop that would drain the prefetched grid loads and the
// streamed stores of every wave at every tile, i.e. serialise HBM latency with the matrix core.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
using f32x16 = __attribute__((ext_vector_type(16))) float;

#ifndef APMF_MF_WAVES
#define APMF_MF_WAVES 3   // waves per SIMD the register budget is held to (<= 168 VGPRs)
#endif
template <int D, bool VAR, bool RED, bool MEAN>
__global__ __launch_bounds__(256, APMF_MF_WAVES) void k_mn_grid_mfma(const GridDims a, const float *__restrict__ p_cross,
                                                      const float *__restrict__ p_rowT,
                                                      const float *__restrict__ p_colpack,
                                                      const float *__restrict__ p_state,
                                                      const uint16_t *__restrict__ p_maskT,
                                                      float *__restrict__ p_var,
                                                      float *__restrict__ p_pmean,
                                                      double *__restrict__ part_sum,
                                                      float *__restrict__ part_val,
                                                      long long *__restrict__ part_idx)
{
    static_assert(D % 8 == 0 && D <= 64, "D/2 must be a multiple of 4");
    constexpr int KH = D / 2;
    constexpr int AROWS = MEAN ? 2 * D + 2 : D + 2;   // staged rows of a user tile
    constexpr int TSTRIDE = (2 * D + 2) * 32;         // floats per user tile in rowT
    constexpr int NF4 = AROWS * 8;                    // float4 per staged tile
    constexpr int NLD = (NF4 + 255) / 256;            // float4 per thread per tile
    static_assert(NLD <= 5, "5 staging fragments");
    __shared__ __attribute__((aligned(16))) float sA[2][NLD * 1024];

    const int lane = threadIdx.x & 63, wave = wave_id_uniform();
    const int h = lane >> 5, c = lane & 31;
    const int per_state = a.nblk_x * a.nblk_y;
    const int L = blockIdx.x;
    const int full_grp = (a.B >> 3) * per_state * 8;
    int b, tile;
    if (L < full_grp) {
        const int xcd = L & 7, q = L >> 3;
        b = (q / per_state) * 8 + xcd;
        tile = q % per_state;
    } else {
        const int q = L - full_grp;
        b = (a.B & ~7) + q / per_state;
        tile = q % per_state;
    }
    const int bx = tile % a.nblk_x, by = tile / a.nblk_x;
    const int ld = a.ld, n = a.n, m = a.m;
    const int ntile = (n + 31) >> 5;
    const int T0 = by * a.csteps;
    const int nT = min(a.csteps, ntile - T0);
    const int col0 = bx * MF_BCOLS + wave * 32;       // wave-uniform; ld is a multiple of 64
    const bool cok = col0 < ld;                       // a whole wave is in or out
    const int col = col0 + c;
    const float t2 = p_state[2 * b], trO = p_state[2 * b + 1];
    static_assert(VAR || RED || MEAN, "nothing to compute");

    // B operand: this lane's item, k in [h KH, (h+1) KH); then the per-item scalars
    float bw[KH];
    float qq = 0.f, sjj = 0.f;
    {
        const float *cp = p_colpack + (size_t)b * (D + 2) * ld + (cok ? col : 0);
#pragma unroll
        for (int s = 0; s < KH; s++) bw[s] = cok ? cp[(size_t)(h * KH + s) * ld] : 0.f;
        if (cok) { qq = cp[(size_t)D * ld]; sjj = cp[(size_t)(D + 1) * ld]; }
    }
    const float cA = fmaf(t2, sjj, qq);
    const float ht2 = .5f * t2;

    // staging of a user tile: NLD float4 per thread (plain locals: never an indexed stack
    // object); fragments past the tile repeat its last float4 and land in unread LDS padding
    const float *rowT_b = p_rowT + (size_t)b * ntile * TSTRIDE;
    float4 st0, st1, st2, st3, st4;
    st0 = st1 = st2 = st3 = st4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define MF_FETCH1(k, fsrc) \
    if (NLD > k) st##k = *reinterpret_cast<const float4 *>((fsrc) + 4 * min(k * 256 + (int)threadIdx.x, NF4 - 1));
#define MF_FETCH(T)                                                                             \
    {                                                                                            \
        const float *fsrc = rowT_b + (size_t)(T0 + (T)) * TSTRIDE;                               \
        MF_FETCH1(0, fsrc) MF_FETCH1(1, fsrc) MF_FETCH1(2, fsrc) MF_FETCH1(3, fsrc) MF_FETCH1(4, fsrc) \
    }
#define MF_COMMIT1(k, buf) \
    if (NLD > k) *reinterpret_cast<float4 *>(&sA[buf][(k * 256 + threadIdx.x) * 4]) = st##k;
#define MF_COMMIT(buf)                                                                          \
    { MF_COMMIT1(0, buf) MF_COMMIT1(1, buf) MF_COMMIT1(2, buf) MF_COMMIT1(3, buf) MF_COMMIT1(4, buf) }

    // The streamed operand: register i of a tile is row 32 T + RL(i) + 4h.  The row part of the
    // address is wave-uniform (scalar base), the lane part one 32-bit offset.  Only the last
    // tile of a state needs row guards.
#define MF_RL(i) (((i) & 3) + 8 * ((i) >> 2))
    const size_t state_base = (size_t)b * n * ld + col0;      // wave-uniform
    const unsigned lane_off = (unsigned)(4 * h * ld + c);
    // FAST (compile-time): a full 32-row tile of an in-range wave - no guards and no branches,
    // so every path issues the same memory operations and the wait-count bookkeeping of the
    // compiler stays exact (a conditional load anywhere makes it wait for everything in flight).
#define MF_LOAD_TILE(dst, dmb, T, FAST)                                                         \
    {                                                                                            \
        const int lr0 = (T0 + (T)) * 32;                                                         \
        const int lrem = min(32, n - lr0);                                                       \
        const float *lsrc = p_cross + state_base + (size_t)lr0 * ld;                             \
        if (FAST) {                                                                              \
            _Pragma("unroll") for (int i = 0; i < 16; i++)                                       \
                dst[i] = __builtin_nontemporal_load(lsrc + (size_t)MF_RL(i) * ld + lane_off);    \
        } else if (!cok) {                                                                       \
            _Pragma("unroll") for (int i = 0; i < 16; i++) dst[i] = 0.f;                         \
        } else {                                                                                 \
            _Pragma("unroll") for (int i = 0; i < 16; i++)                                       \
                dst[i] = (MF_RL(i) + 4 * h < lrem)                                               \
                             ? __builtin_nontemporal_load(lsrc + (size_t)MF_RL(i) * ld + lane_off) \
                             : 0.f;                                                              \
        }                                                                                        \
        if (RED) {                                                                               \
            /* candidate bits, pre-shifted by 4h: bit RL(i) <-> accumulator register i */        \
            const size_t g0 = (size_t)(lr0 >> 4);                                                \
            if (FAST) {                                                                          \
                const unsigned lo = p_maskT[g0 * ld + col], hi = p_maskT[(g0 + 1) * ld + col];   \
                dmb = col < m ? ((lo | (hi << 16)) >> (4 * h)) : 0u;                             \
            } else {                                                                             \
                unsigned w = 0u;                                                                 \
                if (cok && lrem > 0) {                                                           \
                    const unsigned lo = p_maskT[g0 * ld + col];                                  \
                    const unsigned hi = (lrem > 16) ? p_maskT[(g0 + 1) * ld + col] : 0u;         \
                    w = lo | (hi << 16);                                                         \
                }                                                                                \
                const unsigned rowbits = lrem >= 32 ? 0xFFFFFFFFu : (lrem > 0 ? (1u << lrem) - 1u : 0u); \
                dmb = (cok && col < m) ? ((w & rowbits) >> (4 * h)) : 0u;                        \
            }                                                                                    \
        }                                                                                        \
    }

    double sum = 0.;
    const float sgn = a.want < 0 ? -1.f : 1.f;          // arg-extremum on key = sgn * value
    float bestkey = -__builtin_inff();
    int bestcode = -1;

    // one tile: prefetch the next tile's operands (TN), run the matrix core on this one, finish it
#define MF_TILE(T, TN, cur, curmb, nxt, nxtmb, buf, FAST)                                       \
    {                                                                                            \
        MF_FETCH(min((T) + 1, nT - 1));                                                          \
        __builtin_amdgcn_sched_barrier(0);   /* staged operands first: they are needed first */  \
        MF_LOAD_TILE(nxt, nxtmb, TN, FAST);                                                      \
        /* keep the loads up here: sunk next to their uses they would be waited for with */      \
        /* everything younger already in flight, i.e. a full drain per tile              */      \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        f32x16 acc, accm;                                                                        \
        _Pragma("unroll") for (int i = 0; i < 16; i++) { acc[i] = 0.f; accm[i] = 0.f; }          \
        const float *at = &sA[buf][(h * KH) * 32 + c];                                           \
        /* A operands in batches of 8 LDS reads ahead of their 8 MFMAs (one wait per batch) */   \
        _Pragma("unroll") for (int s0 = 0; s0 < KH; s0 += 8) {                                   \
            float av[8], avm[8];                                                                 \
            _Pragma("unroll") for (int q = 0; q < 8; q++) {                                      \
                if (s0 + q < KH) {                                                               \
                    av[q] = at[(s0 + q) * 32];                                                   \
                    if (MEAN) avm[q] = at[(D + 2 + s0 + q) * 32];                                \
                }                                                                                \
            }                                                                                    \
            _Pragma("unroll") for (int q = 0; q < 8; q++) {                                      \
                if (s0 + q < KH) {                                                               \
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bw[s0 + q], acc, 0, 0, 0); \
                    if (MEAN) accm = __builtin_amdgcn_mfma_f32_32x32x2f32(avm[q], bw[s0 + q], accm, 0, 0, 0); \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
        /* the other LDS buffer was last read before the previous barrier: the staged next  */   \
        /* tile can land now, while only this tile's younger grid loads are still in flight */   \
        MF_COMMIT((buf) ^ 1);                                                                    \
        const int r0 = (T0 + (T)) * 32;                                                          \
        const int rem = min(32, n - r0);                                                         \
        const float *prow = &sA[buf][D * 32 + 4 * h];                                            \
        /* Var = s_ii (t2 s_jj + q_j) + p_i s_jj + 2 x (t2 x / 2 + G) */                          \
        _Pragma("unroll") for (int i = 0; i < 16; i++) {                                         \
            const float pi = prow[MF_RL(i)], sii = prow[32 + MF_RL(i)];                          \
            const float x = cur[i];                                                              \
            const float u = fmaf(ht2, x, acc[i]);                                                \
            acc[i] = fmaf(sii, cA, fmaf(x + x, u, pi * sjj));                                    \
            if (MEAN) accm[i] = fmaf(x, trO, accm[i]);                                           \
        }                                                                                        \
        float *vdst = p_var + state_base + (size_t)r0 * ld;                                      \
        float *mdst = p_pmean + state_base + (size_t)r0 * ld;                                    \
        if (FAST) {                                                                              \
            _Pragma("unroll") for (int i = 0; i < 16; i++) {                                     \
                if (VAR) __builtin_nontemporal_store(acc[i], vdst + (size_t)MF_RL(i) * ld + lane_off); \
                if (MEAN) __builtin_nontemporal_store(accm[i], mdst + (size_t)MF_RL(i) * ld + lane_off); \
            }                                                                                    \
        } else if (cok) {                                                                        \
            _Pragma("unroll") for (int i = 0; i < 16; i++) {                                     \
                if (MF_RL(i) + 4 * h < rem) {                                                    \
                    if (VAR) __builtin_nontemporal_store(acc[i], vdst + (size_t)MF_RL(i) * ld + lane_off); \
                    if (MEAN) __builtin_nontemporal_store(accm[i], mdst + (size_t)MF_RL(i) * ld + lane_off); \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
        if (RED) {                                                                               \
            float tsum = 0.f;                                                                    \
            _Pragma("unroll") for (int i = 0; i < 16; i++) {                                     \
                const int sel = __builtin_amdgcn_sbfe((int)curmb, MF_RL(i), 1);   /* 0 or -1 */  \
                if (a.want_sum) tsum += __int_as_float(__float_as_int(acc[i]) & sel);            \
                if (a.want) {                                                                    \
                    const float key = sgn * acc[i];                                              \
                    const bool better = sel && key > bestkey;                                    \
                    bestkey = better ? key : bestkey;                                            \
                    bestcode = better ? (T) * 16 + i : bestcode;                                 \
                }                                                                                \
            }                                                                                    \
            sum += (double)tsum;                                                                 \
        }                                                                                        \
        lds_barrier();                                                                           \
    }

    // tiles [0, nfull) of this walk are full 32-row tiles of an in-range wave: branch-free pairs
    // with ping-pong register sets; what is left (the state's last, partial tile; every tile of
    // an out-of-range wave, which only has to keep the barriers) goes through the guarded form
    const int nfull = cok ? max(0, min(nT, (n >> 5) - T0)) : 0;
    float svA[16], svB[16];
    unsigned mbA = 0u, mbB = 0u;
    MF_FETCH(0);
    MF_COMMIT(0);
    __syncthreads();
    int T = 0;
    if (nfull > 0) {
        MF_LOAD_TILE(svA, mbA, 0, 1);
#pragma unroll 1
        for (; T + 1 < nfull; T += 2) {
            MF_TILE(T, T + 1, svA, mbA, svB, mbB, 0, 1);
            MF_TILE(T + 1, min(T + 2, nfull - 1), svB, mbB, svA, mbA, 1, 1);
        }
        if (T < nfull) {
            MF_TILE(T, T, svA, mbA, svB, mbB, 0, 1);
            T++;
        }
    }
    if (T < nT) {
        MF_LOAD_TILE(svA, mbA, T, 0);
#pragma unroll 1
        for (; T < nT; T++) {
            const int sbuf = T & 1;
            MF_TILE(T, min(T + 1, nT - 1), svA, mbA, svB, mbB, sbuf, 0);
#pragma unroll
            for (int i = 0; i < 16; i++) svA[i] = svB[i];
            mbA = mbB;
        }
    }
    float bestv = sgn * bestkey;
    long long besti = -1;
    if (bestcode >= 0)
        besti = (long long)((T0 + (bestcode >> 4)) * 32 + MF_RL(bestcode & 15) + 4 * h) * m + col;
#undef MF_TILE
#undef MF_FETCH
#undef MF_FETCH1
#undef MF_COMMIT
#undef MF_COMMIT1
#undef MF_LOAD_TILE
#undef MF_RL

    if (RED) {
        __shared__ double sh_sum[WAVES];
        __shared__ float sh_val[WAVES];
        __shared__ long long sh_idx[WAVES];
        sum = wave_sum(sum);
        if (a.want) wave_argext(bestv, besti, a.want);
        if (lane == 0) { sh_sum[wave] = sum; sh_val[wave] = bestv; sh_idx[wave] = besti; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double ts = sh_sum[0];
            float bv = sh_val[0];
            long long bi = sh_idx[0];
            for (int w = 1; w < WAVES; w++) {
                ts += sh_sum[w];
                if (a.want) argext_merge(bv, bi, sh_val[w], sh_idx[w], a.want);
            }
            const size_t slot = (size_t)b * per_state + tile;
            part_sum[slot] = ts;
            part_val[slot] = bv;
            part_idx[slot] = bi;
        }
    }
}

template <int D>
void launch_grid_mfma(const GridArgs<float> &a, hipStream_t st)
{
    const bool red = a.g.want != 0 || a.g.want_sum, mean = a.pmean != nullptr, var = a.var != nullptr;
    dim3 grid((unsigned)((long long)a.g.B * a.g.nblk_x * a.g.nblk_y));
#define APMF_MF(V, R, M)                                                                           \
    hipLaunchKernelGGL((k_mn_grid_mfma<D, V, R, M>), grid, dim3(256), 0, st, a.g, a.cross, a.rowpack, \
                       a.colpack, a.state, a.maskT, a.var, a.pmean, a.part_sum, a.part_val, a.part_idx)
    if (var) {
        if (red && mean) APMF_MF(true, true, true);
        else if (red) APMF_MF(true, true, false);
        else if (mean) APMF_MF(true, false, true);
        else APMF_MF(true, false, false);
    } else {
        if (red && mean) APMF_MF(false, true, true);
        else if (red) APMF_MF(false, true, false);
        else APMF_MF(false, false, true);
    }
#undef APMF_MF
}

// one block per state: fixed-order combine of the per-block partials
template <typename T>
__global__ __launch_bounds__(256) void k_grid_finish(int nblk, int want, const double *__restrict__ part_sum,
                                                     const T *__restrict__ part_val,
                                                     const long long *__restrict__ part_idx,
                                                     double *__restrict__ out_sum, T *__restrict__ out_val,
                                                     int64_t *__restrict__ out_idx)
{
    const int b = blockIdx.x;
    __shared__ double sh_sum[256];
    __shared__ T sh_val[256];
    __shared__ long long sh_idx[256];
    double s = 0;
    T bv = 0;
    long long bi = -1;
    for (int t = threadIdx.x; t < nblk; t += 256) {
        s += part_sum[(size_t)b * nblk + t];
        if (want) argext_merge(bv, bi, part_val[(size_t)b * nblk + t], part_idx[(size_t)b * nblk + t], want);
    }
    sh_sum[threadIdx.x] = s; sh_val[threadIdx.x] = bv; sh_idx[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sh_sum[threadIdx.x] += sh_sum[threadIdx.x + o];
            if (want) {
                T v = sh_val[threadIdx.x];
                long long i = sh_idx[threadIdx.x];
                argext_merge(v, i, sh_val[threadIdx.x + o], sh_idx[threadIdx.x + o], want);
                sh_val[threadIdx.x] = v; sh_idx[threadIdx.x] = i;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (out_sum) out_sum[b] = sh_sum[0];
        if (out_val) out_val[b] = sh_val[0];
        if (out_idx) out_idx[b] = (int64_t)sh_idx[0];
    }
}

template <typename T, int D, int VEC, bool RED, bool MEAN>
void launch_grid_v(const GridArgs<T> &a, hipStream_t st);

template <typename T, int D, int VEC>
void launch_grid(const GridArgs<T> &a, hipStream_t st)
{
    const bool red = a.g.want != 0 || a.g.want_sum, mean = a.pmean != nullptr;
    if (red && mean) launch_grid_v<T, D, VEC, true, true>(a, st);
    else if (red) launch_grid_v<T, D, VEC, true, false>(a, st);
    else if (mean) launch_grid_v<T, D, VEC, false, true>(a, st);
    else launch_grid_v<T, D, VEC, false, false>(a, st);
}

template <typename T, int D, int VEC, bool RED, bool MEAN>
void launch_grid_v(const GridArgs<T> &a, hipStream_t st)
{
    const int per_state = a.g.nblk_x * a.g.nblk_y;
    // full groups of 8 states are dealt one state per XCD; the remaining states tile by tile
    dim3 grid((unsigned)((long long)a.g.B * per_state));
    hipLaunchKernelGGL((k_mn_grid<T, D, VEC, RED, MEAN>), grid, dim3(256), 0, st, a.g, a.cross, a.rowpack,
                       a.colpack, a.state, a.maskT, a.var, a.pmean, a.part_sum, a.part_val,
                       a.part_idx);
}

// padded latent dimension and vector width used for a runtime d
template <typename T> struct Plan { int D, VEC; };
template <typename T>
Plan<T> plan_for(int d)
{
    static const int steps[] = {2, 4, 6, 8, 10, 12, 16, 24, 32, 48, 64};
    Plan<T> p = {0, 0};
    for (int s : steps)
        if (d <= s) { p.D = s; break; }
    const bool f32 = sizeof(T) == 4;
    if (p.D <= 16) p.VEC = f32 ? 4 : 2;
    else if (p.D <= 32) p.VEC = f32 ? 2 : 1;
    else p.VEC = 1;
    return p;
}

template <typename T>
int dispatch_grid(const GridArgs<T> &a, int D, hipStream_t st);

template <>
int dispatch_grid<float>(const GridArgs<float> &a, int D, hipStream_t st)
{
    switch (D) {
    case 2: launch_grid<float, 2, 4>(a, st); break;
    case 4: launch_grid<float, 4, 4>(a, st); break;
    case 6: launch_grid<float, 6, 4>(a, st); break;
    case 8: launch_grid<float, 8, 4>(a, st); break;
    case 10: launch_grid<float, 10, 4>(a, st); break;
    case 12: launch_grid<float, 12, 4>(a, st); break;
    case 16: if (a.g.csteps) launch_grid_mfma<16>(a, st); else launch_grid<float, 16, 4>(a, st); break;
    case 24: if (a.g.csteps) launch_grid_mfma<24>(a, st); else launch_grid<float, 24, 2>(a, st); break;
    case 32: if (a.g.csteps) launch_grid_mfma<32>(a, st); else launch_grid<float, 32, 2>(a, st); break;
    case 48: if (a.g.csteps) launch_grid_mfma<48>(a, st); else launch_grid<float, 48, 1>(a, st); break;
    case 64: if (a.g.csteps) launch_grid_mfma<64>(a, st); else launch_grid<float, 64, 1>(a, st); break;
    default: return APMF_EINVAL;
    }
    return APMF_OK;
}

template <>
int dispatch_grid<double>(const GridArgs<double> &a, int D, hipStream_t st)
{
    switch (D) {
    case 2: launch_grid<double, 2, 2>(a, st); break;
    case 4: launch_grid<double, 4, 2>(a, st); break;
    case 6: launch_grid<double, 6, 2>(a, st); break;
    case 8: launch_grid<double, 8, 2>(a, st); break;
    case 10: launch_grid<double, 10, 2>(a, st); break;
    case 12: launch_grid<double, 12, 2>(a, st); break;
    case 16: launch_grid<double, 16, 2>(a, st); break;
    case 24: launch_grid<double, 24, 1>(a, st); break;
    case 32: launch_grid<double, 32, 1>(a, st); break;
    case 48: launch_grid<double, 48, 1>(a, st); break;
    case 64: launch_grid<double, 64, 1>(a, st); break;
    default: return APMF_EINVAL;
    }
    return APMF_OK;
}

template <typename T>
int mn_score_grid(apmf_ctx *ctx, int B, int n, int m, int d, int ld, const T *d_mean,
                  const T *d_cross, const T *d_diag, const T *d_omega, const uint8_t *d_mask,
                  T *d_var, T *d_pmean, double *d_sum, int want, T *d_best_val,
                  int64_t *d_best_idx)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    APMF_REQUIRE(B >= 1 && n >= 1 && m >= 1, "bad shape B=%d n=%d m=%d", B, n, m);
    APMF_REQUIRE(d >= 1 && d <= APMF_MAX_D, "latent_d=%d unsupported (1..%d)", d, APMF_MAX_D);
    APMF_REQUIRE(ld >= m && ld % APMF_GRID_ALIGN == 0, "ld=%d must be >= m=%d and a multiple of %d",
                 ld, m, APMF_GRID_ALIGN);
    APMF_REQUIRE(d_mean && d_cross && d_diag && d_omega, "NULL operand");
    APMF_REQUIRE(want >= -1 && want <= 1, "want must be -1, 0 or +1");
    APMF_REQUIRE(want == 0 || (d_best_val && d_best_idx), "want != 0 needs d_best_val/d_best_idx");
    APMF_REQUIRE(((uintptr_t)d_cross % 16) == 0 && (!d_var || ((uintptr_t)d_var % 16) == 0) &&
                     (!d_pmean || ((uintptr_t)d_pmean % 16) == 0),
                 "grids must be 16-byte aligned");
    APMF_REQUIRE(!d_mask || ((uintptr_t)d_mask % 4) == 0, "d_mask must be 4-byte aligned");
    if (!d_var && !d_pmean && !d_sum && want == 0) return APMF_OK;

    const Plan<T> pl = plan_for<T>(d);
    int D = pl.D;
    int rs = round_up(2 * D + 2, 4);
    GridArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.g.B = B; a.g.n = n; a.g.m = m; a.g.d = d; a.g.ld = ld; a.g.rs = rs;
    a.g.nblk_x = (ld + 64 * pl.VEC - 1) / (64 * pl.VEC);
    a.g.nblk_y = (n + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    bool mfma = false;
    // float32 kernel choice: the matrix-core sweep from the rank-16 bucket up, and for sweeps
    // that write no grid (sum / arg-extremum only), where it is ahead at every rank; the
    // streaming VALU kernel otherwise.  APMF_K2_KERNEL=mfma|valu overrides (tuning, tests).
    bool use_mfma = sizeof(T) == 4 && (D >= 16 || (!d_var && !d_pmean));
    if (const char *e = getenv("APMF_K2_KERNEL")) {
        if (!strcmp(e, "mfma")) use_mfma = sizeof(T) == 4;
        else if (!strcmp(e, "valu")) use_mfma = false;
    }
    if (use_mfma) {
        if (D < 16) { D = 16; rs = round_up(2 * D + 2, 4); a.g.rs = rs; }
        // matrix-core sweep: 128-column blocks walking csteps tiles of 32 rows; shorter walks
        // when the grid would otherwise be too small to fill 256 CUs
        mfma = true;
        a.g.nblk_x = (ld + MF_BCOLS - 1) / MF_BCOLS;
        const int ntile = (n + 31) / 32;
        int cs = 16;
        while (cs > 1 && (long long)B * a.g.nblk_x * ((ntile + cs - 1) / cs) < 2048) cs >>= 1;
        a.g.csteps = cs;
        a.g.nblk_y = (ntile + cs - 1) / cs;
    }
    const size_t nblk = (size_t)a.g.nblk_x * a.g.nblk_y;

    // workspace 0: packs; workspace 1: partials
    const size_t rp_el = mfma ? (size_t)B * ((n + 31) / 32) * (2 * D + 2) * 32 : (size_t)B * n * rs, cp_el = (size_t)B * (D + 2) * ld, st_el = (size_t)B * 2;
    void *w0;
    APMF_TRY(apmf_ws(ctx, 0, (rp_el + cp_el + st_el) * sizeof(T) + 64, &w0));
    T *rowpack = (T *)w0;
    T *colpack = rowpack + rp_el;
    T *state = colpack + cp_el;
    void *w1;
    const size_t part_bytes = (size_t)B * nblk * (sizeof(double) + sizeof(T) + sizeof(long long)) + 64;
    APMF_TRY(apmf_ws(ctx, 1, part_bytes, &w1));
    a.part_sum = (double *)w1;
    a.part_idx = (long long *)(a.part_sum + (size_t)B * nblk);
    a.part_val = (T *)(a.part_idx + (size_t)B * nblk);

    const int prep_rows = d <= 16 ? 256 : 64;
    size_t prep_smem = ((size_t)prep_rows * (size_t)(d | 1) + (size_t)d * d) * sizeof(T);   // rows + Omega
    // + the staged row-major user pack, when it fits the default 64 KB
    const size_t stage_bytes = (size_t)prep_rows * (rs + 1) * sizeof(T);
    const int stage = (!mfma && prep_smem + stage_bytes <= 64 * 1024) ? 1 : 0;
    if (stage) prep_smem += stage_bytes;
    const dim3 prep_grid((n + ld + prep_rows - 1) / prep_rows, B);
    if (d <= 4)
        hipLaunchKernelGGL((k_mn_prep<T, 4>), prep_grid, dim3(prep_rows), prep_smem, ctx->stream, B, n, m, d,
                           D, ld, rs, (int)mfma, stage, d_mean, d_diag, d_omega, rowpack, colpack, state);
    else if (d <= 10)
        hipLaunchKernelGGL((k_mn_prep<T, 10>), prep_grid, dim3(prep_rows), prep_smem, ctx->stream, B, n, m,
                           d, D, ld, rs, (int)mfma, stage, d_mean, d_diag, d_omega, rowpack, colpack, state);
    else if (d <= 16)
        hipLaunchKernelGGL((k_mn_prep<T, 16>), prep_grid, dim3(prep_rows), prep_smem, ctx->stream, B, n, m,
                           d, D, ld, rs, (int)mfma, stage, d_mean, d_diag, d_omega, rowpack, colpack, state);
    else
        hipLaunchKernelGGL((k_mn_prep<T, 0>), prep_grid, dim3(prep_rows), prep_smem, ctx->stream, B, n, m, d,
                           D, ld, rs, (int)mfma, stage, d_mean, d_diag, d_omega, rowpack, colpack, state);
    APMF_CHECK_LAUNCH();
    const uint16_t *maskT = nullptr;
    if (d_mask && (want != 0 || d_sum)) {
        const int ngrp = (n + 15) / 16;
        void *w2;
        APMF_TRY(apmf_ws(ctx, 2, (size_t)ngrp * ld * sizeof(uint16_t) + 64, &w2));
        hipLaunchKernelGGL(k_mask_pack, dim3((ld / 4 + 255) / 256, ngrp), dim3(256), 0, ctx->stream, n, ld,
                           d_mask, (uint16_t *)w2);
        APMF_CHECK_LAUNCH();
        maskT = (const uint16_t *)w2;
    } else if (mfma && (want != 0 || d_sum)) {
        // the matrix-core sweep always reads candidate words (no branch in its tile loop):
        // "everything is a candidate" is a table of ones
        const int ngrp = (n + 15) / 16;
        void *w2;
        APMF_TRY(apmf_ws(ctx, 2, (size_t)ngrp * ld * sizeof(uint16_t) + 64, &w2));
        APMF_CHECK_HIP(hipMemsetAsync(w2, 0xFF, (size_t)ngrp * ld * sizeof(uint16_t), ctx->stream));
        maskT = (const uint16_t *)w2;
    }

    a.cross = d_cross; a.rowpack = rowpack; a.colpack = colpack; a.state = state;
    a.maskT = maskT; a.var = d_var; a.pmean = d_pmean; a.g.want = want; a.g.want_sum = d_sum != nullptr;
    int rc;
    {
        ProfScope prof(ctx);
        rc = dispatch_grid<T>(a, D, ctx->stream);
    }
    if (rc != APMF_OK) { apmf_set_error("no kernel for padded latent_d=%d", D); return rc; }
    APMF_CHECK_LAUNCH();
    if (want != 0 || d_sum) {
        hipLaunchKernelGGL((k_grid_finish<T>), dim3(B), dim3(256), 0, ctx->stream, (int)nblk, want,
                           a.part_sum, a.part_val, a.part_idx, d_sum, d_best_val, d_best_idx);
        APMF_CHECK_LAUNCH();
    }
    return APMF_OK;
}

// ---- arbitrary pool of candidates, float64 ------------------------------------------
__global__ __launch_bounds__(256) void k_mn_pool(int n, int m, int d, int ld,
                                                 const double *__restrict__ mean,
                                                 const double *__restrict__ cross,
                                                 const double *__restrict__ diag,
                                                 const double *__restrict__ omega,
                                                 const int32_t *__restrict__ pi,
                                                 const int32_t *__restrict__ pj, int64_t P,
                                                 double *__restrict__ pmean, double *__restrict__ var)
{
    extern __shared__ double sO[];  // d*d mirrored omega, then t2, tr
    for (int t = threadIdx.x; t < d * d; t += blockDim.x) {
        int k = t / d, l = t % d;
        sO[t] = omega[k <= l ? k * d + l : l * d + k];
    }
    __syncthreads();
    double t2 = 0, tr = 0;
    for (int k = 0; k < d; k++) {
        tr += sO[k * d + k];
        for (int l = 0; l < d; l++) t2 += sO[k * d + l] * sO[k * d + l];
    }
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < P; p += stride) {
        const int i = pi[p], j = pj[p];
        if (i < 0 || i >= n || j < 0 || j >= m) {  // never index out of bounds
            if (pmean) pmean[p] = __builtin_nan("");
            if (var) var[p] = __builtin_nan("");
            continue;
        }
        const double *a = mean + (size_t)i * d, *b = mean + (size_t)(n + j) * d;
        const double sij = cross[(size_t)i * ld + j], sii = diag[i], sjj = diag[n + j];
        double aOa = 0, bOb = 0, aOb = 0, ab = 0;
        for (int k = 0; k < d; k++) {
            double oa = 0, ob = 0;
            for (int l = 0; l < d; l++) { oa += sO[k * d + l] * a[l]; ob += sO[k * d + l] * b[l]; }
            aOa += a[k] * oa; bOb += b[k] * ob; aOb += b[k] * oa; ab += a[k] * b[k];
        }
        if (var) var[p] = (sii * sjj + sij * sij) * t2 + sjj * aOa + sii * bOb + 2 * sij * aOb;
        if (pmean) pmean[p] = ab + sij * tr;
    }
}

// ---- near-extremum collection after an f32 sweep -------------------------------------
__global__ __launch_bounds__(256) void k_collect_near(int n, int m, int ld, const float *__restrict__ var,
                                                      const uint8_t *__restrict__ mask, int want,
                                                      float best, float tol, int32_t *__restrict__ pi,
                                                      int32_t *__restrict__ pj, int cap,
                                                      int *__restrict__ count)
{
    // rows over blocks, 4 adjacent columns per thread (16-byte score load, 4-byte mask load);
    // near-ties are rare, so the atomic append is off the common path
    const float lo = best - tol, hi = best + tol;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const float *vrow = var + (size_t)i * ld;
        const uint8_t *mrow = mask ? mask + (size_t)i * ld : nullptr;
        for (int j = threadIdx.x * 4; j < ld; j += 1024) {
            const float4 v4 = *reinterpret_cast<const float4 *>(vrow + j);
            const unsigned mk = mrow ? *reinterpret_cast<const unsigned *>(mrow + j) : 0x01010101u;
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const bool near = want > 0 ? v[c] >= lo : v[c] <= hi;
                if (near && j + c < m && ((mk >> (8 * c)) & 0xFFu)) {
                    const int slot = atomicAdd(count, 1);
                    if (slot < cap) { pi[slot] = i; pj[slot] = j + c; }
                }
            }
        }
    }
}

// every candidate whose float64 score EQUALS the extremum (exact ties), for the reference's
// first-in-pool-order tie rule
__global__ __launch_bounds__(256) void k_collect_eq(int n, int m, int ld, const double *__restrict__ var,
                                                    const uint8_t *__restrict__ mask, double best,
                                                    int32_t *__restrict__ pi, int32_t *__restrict__ pj, int cap,
                                                    int *__restrict__ count)
{
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const double *vrow = var + (size_t)i * ld;
        const uint8_t *mrow = mask ? mask + (size_t)i * ld : nullptr;
        for (int j = threadIdx.x; j < m; j += 256) {
            if (vrow[j] == best && (!mrow || mrow[j])) {
                const int slot = atomicAdd(count, 1);
                if (slot < cap) { pi[slot] = i; pj[slot] = j; }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_gather(const double *__restrict__ grid, int ld,
                                                const int32_t *__restrict__ pi,
                                                const int32_t *__restrict__ pj, int64_t P,
                                                double *__restrict__ out)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < P; p += stride) out[p] = grid[(size_t)pi[p] * ld + pj[p]];
}

// first arg-extremum of a vector in index order; NaNs never win
__global__ __launch_bounds__(256) void k_pool_argext_part(const double *__restrict__ v, int64_t P, int want,
                                                          double *__restrict__ pv, long long *__restrict__ pidx)
{
    double bv = 0;
    long long bi = -1;
    const int64_t per = (P + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = min(lo + per, P);
    for (int64_t t = lo + threadIdx.x; t < hi; t += blockDim.x) {
        const double x = v[t];
        if (bi < 0 ? (x == x) : (want > 0 ? x > bv : x < bv)) { bv = x; bi = t; }
    }
    __shared__ double sv[4];
    __shared__ long long si[4];
    wave_argext(bv, bi, want);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) argext_merge(bv, bi, sv[w], si[w], want);
        pv[blockIdx.x] = bv;
        pidx[blockIdx.x] = bi;
    }
}

__global__ void k_pool_argext_fin(int nb, int want, const double *__restrict__ pv,
                                  const long long *__restrict__ pidx, double *__restrict__ ov,
                                  long long *__restrict__ oi)
{
    double bv = 0;
    long long bi = -1;
    for (int t = 0; t < nb; t++) argext_merge(bv, bi, pv[t], pidx[t], want);
    *ov = bv;
    *oi = bi;
}

} // namespace

extern "C" {

int apmf_mn_score_grid_f32(apmf_ctx *ctx, int B, int n, int m, int d, int ld, const float *d_mean,
                           const float *d_cross, const float *d_diag, const float *d_omega,
                           const uint8_t *d_mask, float *d_var, float *d_pmean, double *d_sum,
                           int want, float *d_best_val, int64_t *d_best_idx)
{
    return mn_score_grid<float>(ctx, B, n, m, d, ld, d_mean, d_cross, d_diag, d_omega, d_mask, d_var,
                                d_pmean, d_sum, want, d_best_val, d_best_idx);
}

int apmf_mn_score_grid_f64(apmf_ctx *ctx, int B, int n, int m, int d, int ld,
                           const double *d_mean, const double *d_cross, const double *d_diag,
                           const double *d_omega, const uint8_t *d_mask, double *d_var,
                           double *d_pmean, double *d_sum, int want, double *d_best_val,
                           int64_t *d_best_idx)
{
    return mn_score_grid<double>(ctx, B, n, m, d, ld, d_mean, d_cross, d_diag, d_omega, d_mask, d_var,
                                 d_pmean, d_sum, want, d_best_val, d_best_idx);
}

int apmf_mn_score_pool_f64(apmf_ctx *ctx, int n, int m, int d, int ld, const double *d_mean,
                           const double *d_cross, const double *d_diag, const double *d_omega,
                           const int32_t *d_pi, const int32_t *d_pj, int64_t P, double *d_pmean,
                           double *d_var)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    APMF_REQUIRE(n >= 1 && m >= 1 && d >= 1 && d <= APMF_MAX_D && ld >= m, "bad shape");
    APMF_REQUIRE(P >= 0, "negative pool size");
    if (P == 0) return APMF_OK;
    APMF_REQUIRE(d_mean && d_cross && d_diag && d_omega && d_pi && d_pj, "NULL operand");
    int blocks = (int)(cdiv64(P, 256) < 2048 ? cdiv64(P, 256) : 2048);
    hipLaunchKernelGGL(k_mn_pool, dim3(blocks), dim3(256), (size_t)d * d * sizeof(double), ctx->stream,
                       n, m, d, ld, d_mean, d_cross, d_diag, d_omega, d_pi, d_pj, P, d_pmean, d_var);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

int apmf_grid_collect_near_f32(apmf_ctx *ctx, int n, int m, int ld, const float *d_var,
                               const uint8_t *d_mask, int want, float best, float rel_tol,
                               int32_t *d_pi, int32_t *d_pj, int cap, int *h_count)
{
    APMF_REQUIRE(ctx && d_var && d_pi && d_pj && h_count, "NULL operand");
    APMF_REQUIRE(want == 1 || want == -1, "want must be +1 or -1");
    APMF_REQUIRE(n >= 1 && m >= 1 && ld >= m && cap >= 1, "bad shape");
    APMF_REQUIRE(ld % 4 == 0 && ((uintptr_t)d_var % 16) == 0 && (!d_mask || ((uintptr_t)d_mask % 4) == 0),
                 "grid pitch must be a multiple of 4 and the grids 16-byte (mask: 4-byte) aligned");
    void *w;
    APMF_TRY(apmf_ws(ctx, 2, 64, &w));
    int *d_count = (int *)w;
    APMF_CHECK_HIP(hipMemsetAsync(d_count, 0, sizeof(int), ctx->stream));
    const float tol = rel_tol * (best < 0 ? -best : best);
    const int blocks = n < 8192 ? n : 8192;
    hipLaunchKernelGGL(k_collect_near, dim3(blocks), dim3(256), 0, ctx->stream, n, m, ld, d_var, d_mask,
                       want, best, tol, d_pi, d_pj, cap, d_count);
    APMF_CHECK_LAUNCH();
    return apmf_d2h(ctx, h_count, d_count, sizeof(int));
}

int apmf_grid_collect_eq_f64(apmf_ctx *ctx, int n, int m, int ld, const double *d_var, const uint8_t *d_mask,
                             double best, int32_t *d_pi, int32_t *d_pj, int cap, int *h_count)
{
    APMF_REQUIRE(ctx && d_var && d_pi && d_pj && h_count, "NULL operand");
    APMF_REQUIRE(n >= 1 && m >= 1 && ld >= m && cap >= 1, "bad shape");
    void *w;
    APMF_TRY(apmf_ws(ctx, 2, 64, &w));
    int *d_count = (int *)w;
    APMF_CHECK_HIP(hipMemsetAsync(d_count, 0, sizeof(int), ctx->stream));
    const int blocks = n < 8192 ? n : 8192;
    hipLaunchKernelGGL(k_collect_eq, dim3(blocks), dim3(256), 0, ctx->stream, n, m, ld, d_var, d_mask, best, d_pi,
                       d_pj, cap, d_count);
    APMF_CHECK_LAUNCH();
    return apmf_d2h(ctx, h_count, d_count, sizeof(int));
}

int apmf_grid_gather_f64(apmf_ctx *ctx, const double *d_grid, int ld, const int32_t *d_pi,
                         const int32_t *d_pj, int64_t P, double *d_out)
{
    APMF_REQUIRE(ctx, "ctx is NULL");
    if (P == 0) return APMF_OK;
    APMF_REQUIRE(d_grid && d_pi && d_pj && d_out && P > 0, "bad operand");
    int blocks = (int)(cdiv64(P, 256) < 2048 ? cdiv64(P, 256) : 2048);
    hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, ctx->stream, d_grid, ld, d_pi, d_pj, P, d_out);
    APMF_CHECK_LAUNCH();
    return APMF_OK;
}

int apmf_pool_argext_f64(apmf_ctx *ctx, const double *d_vals, int64_t P, int want, int64_t *h_pos,
                         double *h_val)
{
    APMF_REQUIRE(ctx && d_vals && h_pos, "NULL operand");
    APMF_REQUIRE(P >= 1, "empty pool");  // mn_active_pmf.py:723-724 raises ValueError
    APMF_REQUIRE(want == 1 || want == -1, "want must be +1 or -1");
    int nb = (int)(cdiv64(P, 4096) < 512 ? cdiv64(P, 4096) : 512);
    void *w;
    APMF_TRY(apmf_ws(ctx, 2, (size_t)(nb + 1) * 16 + 64, &w));
    double *pv = (double *)w;
    long long *pidx = (long long *)(pv + nb + 1);
    hipLaunchKernelGGL(k_pool_argext_part, dim3(nb), dim3(256), 0, ctx->stream, d_vals, P, want, pv, pidx);
    APMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_pool_argext_fin, dim3(1), dim3(1), 0, ctx->stream, nb, want, pv, pidx, pv + nb,
                       pidx + nb);
    APMF_CHECK_LAUNCH();
    double v;
    long long i;
    APMF_TRY(apmf_d2h(ctx, &v, pv + nb, sizeof(double)));
    APMF_TRY(apmf_d2h(ctx, &i, pidx + nb, sizeof(long long)));
    *h_pos = (int64_t)i;
    if (h_val) *h_val = v;
    return APMF_OK;
}

} // extern "C"
