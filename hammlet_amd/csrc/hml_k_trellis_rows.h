// The first pass over the trellis of a weakly compressed sweep, round 3: ROWS.
// Reference: StateSequence<ForwardBackward>::sample, src/StateSequence/ForwardBackward.hpp:67-162 (emission terms :67-84,
// forward recursion :86-123, backward sampling :133-162), Trellis::sample src/Trellis.hpp:61-66, EFD.hpp:23-38.
//
// Same values, bit for bit, as hml_k_trellis_tile (hml_k_trellis.h, round 2: kept as HML_TRELLIS_ROWS=0 and as the
// comparison path of the tests) - every block goes through the same operations in the same order - but a different
// division of labour and a different shape of code.  Round 2's kernel gives the sequential filter one lane per chunk and
// spreads emission terms and candidate maps over (chunk, row) lanes: every batch of 4 rows crosses LDS three times, waits
// at four phase boundaries and reads memory in 16- and 32-byte pieces a chunk length apart (3.9 times its algorithmic
// bytes); its 6000 instructions of straight-line code with every rare case inlined spill a hundred scalar registers.
// Measured on MI355X (tools/valu_bench.hip): a wavefront issues a plain 32-bit VOP2 instruction (v_mul_f32, v_add_f32,
// v_xor_b32) in 2 cycles, everything three-operand, 64-bit or double (v_fma_f32, v_min3_f32, v_mul_f64, v_mul_lo_u32,
// v_cvt_f64_f32, v_pk_mul_f32) in 4, v_rcp_f32 in 8 - the pass is bound by vector issue, so what counts is the number and
// the kind of instructions per block.  Here
//   * a lane keeps its chunk for EVERYTHING that is arithmetic: emission terms, filter step, rescaled row, candidate map
//     and chunk map of a row are one piece of code on registers, rows follow each other without a barrier;
//   * LDS is only the transposer between memory order and lane order: a batch is R = 16 rows of the wavefront's 64
//     chunks; 16 consecutive lanes fetch 16 consecutive blocks of one chunk - 64 contiguous bytes of `starts`, whole
//     128-byte lines of the integral array where blocks are short - reduce them to (N, Sx, Sxx) and park the 12 bytes in
//     the chunk's row of the tile; after the batch the same lanes write the statistics (for the count pass) and the
//     32-bit candidate maps back in 128- and 64-byte runs.  All loads of half a batch are issued before the first use;
//   * the hot path holds only the common case of every step: the inner product through the double reciprocal (exact
//     unless flagged), expf for arguments <= 0, the float screen of the categorical draws in count form.  What the
//     common case cannot decide (a product within 4 ulp of a rounding midpoint, a draw within 2^-17 of a boundary, a
//     negative or non-finite value, a block across a cell boundary of the integral array) sets a flag, and ONE branch
//     per row goes to the literal forms of hml_k_forward.h / hml_k_backward.h;
//   * a workgroup is four such wavefronts, each with a tile of its own (they share the read-only tables), and nothing
//     but compiler barriers between the phases: LDS instructions of one wavefront execute in order.
// 12.6 KB of LDS per wavefront: three wavefronts per SIMD with 168 registers each.
#ifndef HML_K_TRELLIS_ROWS_H
#define HML_K_TRELLIS_ROWS_H

#include "hml_k_trellis.h"

#ifndef HML_TR2_R
#define HML_TR2_R 16        // rows per batch
#endif
#ifndef HML_TR2_WPE
#define HML_TR2_WPE 3       // wavefronts per SIMD the register allocation aims at (up to 8 states)
#endif
#define HML_TR2_WAVES 4     // wavefronts per workgroup
template <int K>
struct hml_tr2 {
    static constexpr int SLOTW = (K <= 8) ? 3 : 4;              // words per block in the tile: {N, then the map} {high half of a 64-bit map} Sx Sxx
    static constexpr int PITCH = HML_TR2_R * SLOTW + 1;         // odd: the rows of 64 chunks fall into different banks
    static constexpr int SX = SLOTW - 2;
};

// the phases of one wavefront hand data to each other through its own tile: LDS instructions of a wavefront execute in
// order, so all that is needed is that the compiler keeps them in order too
__device__ __forceinline__ void hml_wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// x mod 65535 folded to [0, 65535] (65535 stands for 0): 2^16 = 1 (mod 65535), so the two halves of x may be added
__device__ __forceinline__ uint32_t hml_fold_cell(uint32_t x) {
    const uint32_t f = (x & 0xffffu) + (x >> 16);
    return (f & 0xffffu) + (f >> 16);
}

// (Sx, Sxx) of block [st, en) from the integral-array entries a = ia[st], z = ia[en] when no cell boundary lies strictly
// inside the block - hml_block_stats_one with its loop empty: Kahan-adding one term to (0, 0) gives 0 + (v - 0), the
// subtraction aggregator the same (IntegralArray.hpp:104-124, KahanAggregator.hpp:26-45).  `inside` reports a boundary
// inside the block (the caller then takes hml_block_stats_one).
__device__ __forceinline__ void hml_tr2_stats(uint32_t st, uint32_t en, const float2 a, const float2 z, float& sx, float& sq, bool& inside) {
    const uint32_t ge = hml_fold_cell(en), gs = hml_fold_cell(st);
    const bool end_on_cell = (ge == 65535u) || (ge == 0u);           // en % 65535 == 0
    const uint32_t rs = (gs == 65535u) ? 0u : gs;                    // st % 65535
    inside = (en - st) + rs > 65535u;                                // the first boundary behind st lies before en
    const float ps = 0.0f + (a.x - 0.0f), pq = 0.0f + (a.y - 0.0f);
    const float ns = end_on_cell ? 0.0f : 0.0f + (z.x - 0.0f), nq = end_on_cell ? 0.0f : 0.0f + (z.y - 0.0f);
    sx = ps - ns;
    sq = pq - nq;
}

// what the rows need of the model: a copy in LDS (read by every lane at the same address - a broadcast; as scalar
// registers these 30 values, the transition matrix and the kernel's pointers overflowed the 102 there are, and the row
// loop spent 60 v_readlane_b32 per row reloading them)
template <int K>
struct hml_tr2_params {
    double mu2[K];     // 2.0 * (double)mu: exact, the first product of hml_inner_product
    double rvar[K];    // 1 / (2 var)
    float logN[K], logA[K];
};
template <int K>
__device__ __forceinline__ void hml_tr2_params_fill(hml_tr2_params<K>& p, const hml_model* __restrict__ mdl_ro, int tid) {
    if (tid < K) {
        p.mu2[tid] = 2.0 * (double)mdl_ro->mu[tid];
        p.rvar[tid] = mdl_ro->rvar2[tid];
        p.logN[tid] = mdl_ro->logN[tid];
        p.logA[tid] = mdl_ro->self_trans ? mdl_ro->logA[tid] : 0.0f;   // (no self-transition term: (N - 1) * 0 = +0 leaves E as it is)
    }
}

// e^x for x <= 0 - all the rows ever ask for (x = E_s - max E): hml_expf_tab with its case analysis folded.  Below
// -0x1.9fe368p6 (and for -inf) hml_expf_tab answers 0; clamped to -104 the arithmetic gives e^-104 = 6.8e-46, less than half
// the smallest float, which rounds to the same 0.  NOT for a NaN (the clamp would swallow it): the caller looks for one
// among the arguments of a block and takes hml_expf_tab then.
__device__ __forceinline__ float hml_tr2_expf_nonpos(float x, const uint64_t* tab) {
    const float xc = __builtin_amdgcn_fmed3f(x, -104.0f, 0.0f);   // max(x, -104) for x <= 0 in one instruction (no canonicalising copy)
    const double xd = (double)xc;
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0;
    const double Shift = 0x1.8p52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    const double C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    const double C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    double z = InvLn2N * xd;
    double kd = z + Shift;
    const uint32_t ki = (uint32_t)hml_d2u(kd);   // (the table index and the exponent adjustment only take its low 17 bits)
    kd = kd - Shift;
    const double r = z - kd;
    // t = tab + (ki << 47): the shifted word has nothing below bit 47, so only the high half moves (as a register pair:
    // composing the 64-bit integer first costs four copies and a 64-bit add)
    typedef uint32_t hml_u32x2 __attribute__((ext_vector_type(2)));
    hml_u32x2 tw = reinterpret_cast<const hml_u32x2*>(tab)[ki & 31u];
    tw.y += ki << 15;
    const double s = __builtin_bit_cast(double, tw);
    z = C0 * r + C1;
    const double r2 = r * r;
    double y = C2 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    return (float)y;
}

// E_s of one block through the double reciprocal (hml_inner_product's common case).  Returns true when some state's
// product may round differently from the quotient, or leaves the range in which it provably does not: the caller then
// takes hml_tr2_energies_literal for the block (same values wherever this function is exact).
template <int K>
__device__ __forceinline__ bool hml_tr2_energies(const hml_tr2_params<K>& p, bool self, float sx, float sq, float N, float (&E)[K]) {
    const double sxd = (double)sx, sqd = (double)sq;
    const float N1 = N - 1.0f;
    // per state: the distance of the double's low 29 bits from a float midpoint, and of its exponent from the safe range;
    // their minimum / maximum over the states are tested once (one comparison each instead of three per state)
    uint32_t near_min = 0xffffffffu, ex_max = 0u;
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const double num = p.mu2[s] * sxd - sqd;
        const double ipd = num * p.rvar[s];
        const uint64_t bits = hml_d2u(ipd);
        const uint32_t low = ((uint32_t)bits & 0x1fffffffu) - 0x0ffffffcu;             // <= 8: within 4 ulp of a float midpoint
        const uint32_t ex = ((uint32_t)(bits >> 32) & 0x7ff00000u) - (923u << 20);      // > 227 << 20 (+ mantissa bits: masked off): |ip| < 2^-100, > 2^127, inf / NaN, or 0
        near_min = (low < near_min) ? low : near_min;
        ex_max = (ex > ex_max) ? ex : ex_max;
        const float ip = (float)ipd;
        float e = (0.0f + ip) - N * p.logN[s];
        e += N1 * p.logA[s];   // (logA = 0 without self transitions: adds +0)
        E[s] = e;
    }
    // (a product that is exactly zero is outside the exponent range but exact: such blocks - zero coverage - are looked at
    // state by state in the rare branch only)
    if (__builtin_expect(near_min <= 8u || ex_max > (227u << 20), 0)) {
        if (near_min <= 8u) return true;
        bool flagged = false;
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const double ipd = (p.mu2[s] * sxd - sqd) * p.rvar[s];
            const uint32_t ex = (uint32_t)(hml_d2u(ipd) >> 52) & 0x7ffu;
            flagged = flagged || (((ex - 923u) > 227u) && ipd != 0.0);
        }
        return flagged;
    }
    return false;
}

// the same terms literally (hml_inner_product divides where it must and "not finite" is raised, EFD.hpp:23-33)
template <int K>
__device__ __forceinline__ void hml_tr2_energies_literal(const hml_model* __restrict__ mdl_ro, hml_model* mdl, bool self, float sx, float sq,
                                                         float N, float (&E)[K]) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const float ip = hml_inner_product(mdl_ro->mu[s], mdl_ro->var[s], mdl_ro->rvar2[s], sx, sq);
        if (!hml_isfinite(ip)) hml_raise(mdl, HML_DEVERR_IP_NOT_FINITE, ip);
        float e = (0.0f + ip) - N * mdl_ro->logN[s];
        if (self) e += (N - 1.0f) * mdl_ro->logA[s];
        E[s] = e;
    }
}

// The candidate map of a row t < B from its rescaled row and the float copy of its uniform: hml_categorical_k_screen for
// every successor state x, in count form - the running sums of non-negative weights do not decrease, so the first i with
// s_i - u sum >= 0 is the number of negative differences.  `unsure` is set when the screen cannot decide some x (a
// difference within 2^-17 sum of zero, a sum that is tiny, not finite or not positive - which covers every row with a
// NaN) or when the row holds a negative sign (the count form and ForwardBackward.hpp:147-149 both need the literal form
// then): the caller takes hml_tre_cand_u, which agrees with this function wherever `unsure` stays clear.
template <int K>
__device__ __forceinline__ typename hml_tre_map<K>::stored hml_tr2_maps(const float (&row)[K], const hml_amat<K>& A, float uf, bool& unsure,
                                                                        float (&total)[K]) {
    typedef typename hml_tre_map<K>::stored map_t;
    map_t map = 0;
    uint32_t sign = 0u;
#pragma unroll
    for (int s = 0; s < K; ++s) sign |= hml_f2u(row[s]);
    // The number of negative differences is the number of set sign bits: each difference shifts its sign into a code word
    // (one v_alignbit instead of a comparison and an add-with-carry), and the count is the word's population count - taken
    // for all successor states at once where K - 1 <= 4 signs fit a nibble (K <= 5), per successor state beyond.
    // The doubts are gathered as NUMBERS, not as booleans (a chain of `bad = bad || ...` over 4 K comparisons is compiled
    // into three or four vector instructions per comparison that build the flags bit by bit): `worst` is the smallest
    // |difference| - margin (a close call somewhere if not positive), `lowest` the smallest sum, `sums` their sum (a NaN or
    // an infinite sum makes it one: minima skip NaNs, sums do not) - three comparisons per row decide.
    float worst = 3.4028234663852886e38f, lowest = 3.4028234663852886e38f, sums = 0.0f;
    uint32_t code_all = 0u;
#pragma unroll
    for (int xx = 0; xx < K; ++xx) {
        const int x = K - 1 - xx;   // (the last successor state first: its nibble ends up on top)
        float s[K];
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < K; ++i) { acc += row[i] * A[i * K + x]; s[i] = acc; }
        total[x] = acc;   // sum_i row_i A(i, x), added in index order: what the next filter step forms from this row (hml_tr2_step)
        const float t = uf * acc;
        const float margin = acc * 7.62939453125e-06f;   // 2^-17
        lowest = __builtin_fminf(lowest, acc);
        sums += acc;
        uint32_t code = (K == 5) ? code_all : 0u;   // (K = 5: four signs per successor state, the nibbles line up by themselves)
        float nearest = 3.4028234663852886e38f;
#pragma unroll
        for (int i = 0; i < K - 1; ++i) {
            const float d = s[i] - t;
            nearest = __builtin_fminf(nearest, __builtin_fabsf(d));
            code = __builtin_amdgcn_alignbit(code, hml_f2u(d), 31);   // (code << 1) | sign(d)
        }
        worst = __builtin_fminf(worst, nearest - margin);
        if (K == 5) code_all = code;
        else if (K < 5) code_all = (code_all << 4) | code;
        else map |= (map_t)(uint32_t)__builtin_popcount(code) << (4 * x);
    }
    if (K <= 5) {
        // population count of every nibble
        uint32_t c = code_all - ((code_all >> 1) & 0x55555555u);
        c = (c & 0x33333333u) + ((c >> 2) & 0x33333333u);
        map = (map_t)c;
    }
    // a negative sign in the row | a close call | a sum that is tiny | a sum that is not finite (2^-100 < sum < FLT_MAX each)
    unsure = ((sign >> 31) != 0u) | !(worst > 0.0f) | !(lowest > 7.888609052210118e-31f) | !(sums < 3.4028234663852886e38f);
    return map;
}

// (f o g) for maps of up to 8 states in 32 bits (hml_map_compose on half the register width)
template <int K>
__device__ __forceinline__ typename hml_tre_map<K>::stored hml_tr2_compose(typename hml_tre_map<K>::stored f, typename hml_tre_map<K>::stored g) {
    typedef typename hml_tre_map<K>::stored map_t;
    map_t r = 0;
#pragma unroll
    for (int x = 0; x < K; ++x) {
        if (sizeof(map_t) == 4) {
            // bit-field extracts with a register offset (v_bfe_u32) instead of shift, mask, shift, mask
            const uint32_t off = (x == 0) ? ((uint32_t)g << 2) & 60u : ((uint32_t)g >> (4 * x - 2)) & 60u;   // 4 g(x)
            r |= (map_t)(__builtin_amdgcn_ubfe((uint32_t)f, off, 4u) << (4 * x));
        } else {
            const uint32_t y = (uint32_t)(g >> (4 * x)) & 15u;
            r |= ((f >> (4u * y)) & (map_t)15) << (4 * x);
        }
    }
    return r;
}

// (hml_tr2_reciprocal / hml_tr2_quotient: the quotients of the filter step through one double reciprocal - hml_k_forward.h)

// one step of the recursion (hml_fwd_step) with the five quotients taken through one reciprocal
// (pred_j = sum_i alpha_i A(i, j) in index order - hml_tr2_predict, or the totals hml_tr2_maps formed from the same row)
template <int K>
__device__ __forceinline__ void hml_tr2_predict(const hml_fwd_ctx<K>& cx, const float (&alpha)[K], float (&pred)[K]) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
        float tt = 0.0f;
#pragma unroll
        for (int i = 0; i < K; ++i) tt += alpha[i] * cx.A[i * K + j];
        pred[j] = tt;
    }
}
template <int K>
__device__ __forceinline__ bool hml_tr2_step(const hml_fwd_ctx<K>& cx, float (&alpha)[K], const float (&e)[K], const float (&pred)[K]) {
    float f[K];
#pragma unroll
    for (int j = 0; j < K; ++j) f[j] = e[j] * pred[j];
    float Z = 0.0f;
#pragma unroll
    for (int j = 0; j < K; ++j) Z += f[j];
    if (__builtin_expect(!(Z > 0.0f) || !(Z < 3.4028234663852886e38f), 0)) {   // 0: the uniform vector; negative, infinite or NaN: whatever the division says
        const bool ok = (Z != 0.0f);
#pragma unroll
        for (int j = 0; j < K; ++j) alpha[j] = ok ? f[j] / Z : cx.invK;
        return !ok;
    }
    const double Zd = (double)Z;
    const double r = hml_tr2_reciprocal(Zd);
#pragma unroll
    for (int j = 0; j < K; ++j) alpha[j] = hml_tr2_quotient(f[j], Zd, r);
    return false;
}

// SHARE (the host's choice for sweeps over nearly uncompressed input, results the same): see `pred` below
template <int K, bool SHARE>
HML_KERNEL __launch_bounds__(64 * HML_TR2_WAVES) __attribute__((amdgpu_waves_per_eu(K <= 8 ? HML_TR2_WPE : 1, K <= 8 ? HML_TR2_WPE : 8)))
void hml_k_trellis_rows(const float2* __restrict__ ia, const uint32_t* __restrict__ starts, hml_model* __restrict__ mdl,
                        const hml_model* __restrict__ mdl_ro, float2* __restrict__ bstat, unsigned long long* __restrict__ cand,
                        unsigned long long* __restrict__ fmap, float* __restrict__ entry, float* __restrict__ exitv,
                        uint32_t* __restrict__ fb_count, float* __restrict__ eprobe, float* __restrict__ aprobe,
                        uint32_t* __restrict__ ckpt, uint32_t L) {
    constexpr int R = HML_TR2_R, SLOTW = hml_tr2<K>::SLOTW, PITCH = hml_tr2<K>::PITCH, SX = hml_tr2<K>::SX;
    typedef typename hml_tre_map<K>::stored map_t;
    static_assert(R % 2 == 0 && HML_TRE_MIN_L % R == 0 && HML_TRE_HALO % R == 0, "row pairs share a Philox block; chunks and warm-ups are whole batches");
    __shared__ uint32_t sm_tile[HML_TR2_WAVES][HML_TRE_NCH * PITCH];
    __shared__ float gtab[HML_TRE_GTAB * K];
    __shared__ uint64_t etab[32];
    __shared__ float sm_A[hml_amat<K>::LDS_FLOATS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* const tile = sm_tile[wave];
    const uint32_t B = mdl_ro->B;
    __shared__ hml_tr2_params<K> p;
    hml_tr2_params_fill<K>(p, mdl_ro, (int)threadIdx.x);
    const bool self = mdl_ro->self_trans != 0;
    const uint32_t Wt = hml_tre_warmup(mdl_ro);
    const int Wr = (int)((Wt + (uint32_t)R - 1u) / (uint32_t)R * (uint32_t)R);
    const unsigned long long epoch = mdl_ro->epoch;
    const hml_key key = mdl_ro->key;
    for (int i = threadIdx.x; i < HML_TRE_GTAB * K; i += 64 * HML_TR2_WAVES) gtab[i] = hml_expf(((float)(i / K + 1) - 1.0f) * mdl_ro->logA[i % K]);   // N = i / K + 1
    if (threadIdx.x < 32) etab[threadIdx.x] = HML_EXP2F_TAB[threadIdx.x];
    hml_amat_fill<K>(sm_A, mdl_ro, (int)threadIdx.x, 64 * HML_TR2_WAVES);
    hml_fwd_ctx<K> cx;
    hml_fwd_ctx_load<K>(cx, mdl_ro, sm_A);
    if (blockIdx.x == 0 && threadIdx.x < K && aprobe) aprobe[threadIdx.x] = mdl_ro->pi[threadIdx.x];
    const uint32_t C = (B + L - 1u) / L;
    const uint32_t n_groups = (C + (uint32_t)HML_TRE_NCH - 1u) / (uint32_t)HML_TRE_NCH;
    const bool probes = eprobe != nullptr || aprobe != nullptr;
    __syncthreads();
    // the (chunk, row) a lane serves while the wavefront moves a batch between memory and the tile: R consecutive lanes per
    // chunk, slot k takes chunks CPS k .. CPS k + CPS - 1
    constexpr int CPS = 64 / R;
    static_assert(CPS * R == 64, "a slot is whole chunks");
    const int sr = lane % R, sc = lane / R;
    for (uint32_t grp = blockIdx.x * (uint32_t)HML_TR2_WAVES + (uint32_t)wave; grp < n_groups; grp += gridDim.x * (uint32_t)HML_TR2_WAVES) {   // wave-uniform
        const uint32_t f0 = grp * (uint32_t)HML_TRE_NCH;
        const uint32_t f = f0 + (uint32_t)lane;   // this lane's own chunk
        const long long first = (long long)f * L;
        const bool active = f < C;
        const long long last = active ? ((first + L < (long long)B) ? first + L : (long long)B) : first;
        const long long ws = (first >= (long long)Wt) ? first - (long long)Wt : 0ll;
        float alpha[K];
#pragma unroll
        for (int s = 0; s < K; ++s) alpha[s] = (ws == 0ll) ? mdl_ro->pi[s] : cx.invK;
        uint32_t nfb = 0u;
        map_t cmap = (map_t)HML_MAP_IDENTITY;
        // The candidate maps of a row and the filter step of the next row both start from sum_i alpha_i A(i, x), added in
        // index order: where the row went into its maps unrescaled (every block of the wavefront's rows a single position -
        // uncompressed input), the step takes the maps' totals instead of forming the K x K products again (SHARE; on
        // compressed input some lane nearly always holds a longer block, and the bookkeeping would only cost).
        float pred[K];
        bool pred_ok = false;   // pred[] holds the totals of this lane's previous row
#pragma unroll
        for (int s = 0; s < K; ++s) pred[s] = 0.0f;
        for (int rel0 = -Wr; rel0 < (int)L; rel0 += R) {   // wave-uniform
            // ---------------- in: 64 chunks x 16 rows -> (N, Sx, Sxx) in the tile; two halves of 8 slots, each with its
            // block starts, then its integral-array gathers, in flight together (no branch in between: slots without a
            // block read block 0)
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                uint32_t st[R / 2], en[R / 2];
                bool have[R / 2];
#pragma unroll
                for (int k = 0; k < R / 2; ++k) {
                    // (32-bit: the first block cf of a chunk fc < C is below B < 2^32; row off < L of the chunk is a block if it
                    // lies inside the warm-up window, at or behind block 0 and before B - the chunk's own end is never nearer)
                    const int c = (half * (R / 2) + k) * CPS + sc;
                    const uint32_t fc = f0 + (uint32_t)c;
                    const uint32_t cf = fc * L;
                    const int off = rel0 + sr;
                    have[k] = fc < C && off >= -(int)Wt && ((off < 0) ? cf >= (uint32_t)(-off) : (uint32_t)off < B - cf);
                    const uint32_t bb = have[k] ? cf + (uint32_t)off : 0u;
                    st[k] = starts[bb];
                    en[k] = starts[bb + 1u];
                }
                float2 a[R / 2], z[R / 2];
#pragma unroll
                for (int k = 0; k < R / 2; ++k) {
                    a[k] = ia[st[k]]; z[k] = ia[en[k]];
                }
#pragma unroll
                for (int k = 0; k < R / 2; ++k) {
                    const int c = (half * (R / 2) + k) * CPS + sc;
                    float sx, sq;
                    bool inside;
                    hml_tr2_stats(st[k], en[k], a[k], z[k], sx, sq, inside);
                    if (inside) hml_block_stats_one(ia, st[k], en[k], sx, sq);   // (a block across a cell boundary: one in 65 535 positions)
                    uint32_t* const w = tile + c * PITCH + sr * SLOTW;
                    w[0] = have[k] ? en[k] - st[k] : 0u;
                    w[SX] = hml_f2u(sx);
                    w[SX + 1] = hml_f2u(sq);
                }
            }
            hml_wave_lds_fence();
            // ---------------- the rows of the batch, every lane on its own chunk
            if (rel0 == 0 && active) {
#pragma unroll
                for (int s = 0; s < K; ++s) entry[(uint64_t)f * K + s] = alpha[s];
            }
            if (active) {
                const uint32_t* const mine = tile + lane * PITCH;
                // rows [r_lo, r_hi) of the batch are blocks of this lane's chunk (its warm-up included); 32-bit from here on
                const long long base = first + rel0;
                const int r_lo = (ws > base) ? (int)((ws - base < (long long)R) ? ws - base : (long long)R) : 0;
                const int r_hi = (last > base) ? (int)((last - base < (long long)R) ? last - base : (long long)R) : 0;
                const uint32_t b0 = (uint32_t)base;   // (block numbers below 2^32; only used where the row is a block)
                const int r_first = (rel0 < -(int)Wt) ? -(int)Wt - rel0 : 0;   // (wave-uniform: the first batch of a warm-up that is no multiple of 16)
                double u_odd = 0.0;
#pragma unroll 1
                for (int r = r_first; r < R; ++r) {
                    const uint32_t b = b0 + (uint32_t)r;
                    double u = u_odd;
                    if (rel0 >= 0 && (r & 1) == 0) {   // blocks 2m, 2m + 1 share Philox block m (D1)
                        // (the key words pass through an empty statement: left alone the compiler keeps the ten round keys of
                        // the schedule - two adds each - in twenty scalar registers across the loop, which it does not have)
                        hml_key kk = key;
                        asm volatile("" : "+s"(kk.k0), "+s"(kk.k1));
                        hml_cat_uniform_pair(kk, epoch, b >> 1, u, u_odd);
                    }
                    if (r < r_lo || r >= r_hi) continue;
                    const uint32_t nb = mine[r * SLOTW];
                    const float sx = hml_u2f(mine[r * SLOTW + SX]), sq = hml_u2f(mine[r * SLOTW + SX + 1]);
                    const float N = (float)nb;
                    float E[K], e[K];
                    if (__builtin_expect(hml_tr2_energies<K>(p, self, sx, sq, N, E), 0)) hml_tr2_energies_literal<K>(mdl_ro, mdl, self, sx, sq, N, E);
                    // std::max in state order from numeric_limits<float>::lowest() (ForwardBackward.hpp:78-81) is the largest term
                    // unless one is a NaN (which the comparison chain lets through and then forgets); v_max3 skips NaNs, and the
                    // NaN shows in the sum below, where the chain is then walked literally
                    float maxE = -3.40282346638528859812e+38f;
#pragma unroll
                    for (int s = 0; s < K; ++s) maxE = __builtin_fmaxf(maxE, E[s]);
                    float xs[K], xsum = 0.0f;
#pragma unroll
                    for (int s = 0; s < K; ++s) { xs[s] = E[s] - maxE; xsum += xs[s]; }   // every term <= 0 or NaN: the sum is a NaN only if a term is
                    if (__builtin_expect(xsum != xsum, 0)) {
                        maxE = -3.40282346638528859812e+38f;
#pragma unroll
                        for (int s = 0; s < K; ++s) maxE = (E[s] < maxE) ? maxE : E[s];
#pragma unroll
                        for (int s = 0; s < K; ++s) e[s] = hml_expf_tab(E[s] - maxE, etab);
                    } else {
#pragma unroll
                        for (int s = 0; s < K; ++s) e[s] = hml_tr2_expf_nonpos(xs[s], etab);
                    }
                    bool fb = false;
                    {
                        if (!SHARE || __ballot(!pred_ok) != 0ull) hml_tr2_predict<K>(cx, alpha, pred);   // (the same values where pred_ok holds)
                        fb = hml_tr2_step<K>(cx, alpha, e, pred);
                        pred_ok = false;
                    }
                    if (rel0 >= 0) {
                        if (fb) nfb++;
                        const uint32_t t = b + 1u;
                        if (__builtin_expect(probes, 0)) {   // (tests: E_s and the unscaled rows)
#pragma unroll
                            for (int s = 0; s < K; ++s) {
                                if (eprobe) eprobe[(uint64_t)b * K + s] = E[s];
                                if (aprobe) aprobe[(uint64_t)t * K + s] = alpha[s];
                            }
                        }
                        float row[K];
#pragma unroll
                        for (int s = 0; s < K; ++s) row[s] = alpha[s];
                        // the reference rescales row t < B after step t + 1 has consumed it (ForwardBackward.hpp:115-119); blocks of
                        // one position have the factor expf(0) = 1, and a wavefront that holds nothing else skips the step
                        bool rescaled = false;   // wave-uniform
                        if (self && __ballot(nb != 1u && t < B) != 0ull) {
                            rescaled = true;
                            if (t < B) {
#pragma unroll
                                for (int s = 0; s < K; ++s)
                                    row[s] = row[s] * ((nb <= (uint32_t)HML_TRE_GTAB) ? gtab[(nb - 1u) * K + s] : hml_expf_tab(((float)nb - 1.0f) * p.logA[s], etab));
                            }
                        }
                        bool unsure = false;
                        map_t cm;
                        {
                            float total[K];
                            cm = hml_tr2_maps<K>(row, cx.A, (float)u, unsure, total);
                            if (SHARE && !rescaled) {
#pragma unroll
                                for (int s = 0; s < K; ++s) pred[s] = total[s];
                                pred_ok = true;
                            }
                        }
                        if (__builtin_expect(unsure || t >= B, 0)) cm = (map_t)hml_tre_cand_u<K>(row, cx.A, mdl, t, B, u);   // (also the last row's constant map)
                        uint32_t* const w = tile + lane * PITCH + r * SLOTW;
                        w[0] = (uint32_t)cm;
                        if (SLOTW == 4) w[1] = (uint32_t)(cm >> 32);
                        cmap = hml_tr2_compose<K>(cmap, cm);
                    }
                }
                // a checkpoint every HML_TRE_CKPT_ROWS rows inside the chunk: the forward vector (and the fallbacks so far) a refit of this
                // chunk compares with - from where it meets them again, bit for bit, the rest of the chunk stands
                // (hml_k_trellis_refit).  Plane-major, the chunk fastest: consecutive lanes write consecutive words.
                if (rel0 >= 0 && ((rel0 + R) & (HML_TRE_CKPT_ROWS - 1)) == 0 && rel0 + R < (int)L) {   // wave-uniform
                    uint32_t* const ck = ckpt + (uint64_t)((uint32_t)(rel0 + R) / (uint32_t)HML_TRE_CKPT_ROWS - 1u) * (uint32_t)(K + 1) * C + f;
#pragma unroll
                    for (int s = 0; s < K; ++s) ck[(uint64_t)s * C] = hml_f2u(alpha[s]);
                    ck[(uint64_t)K * C] = nfb;
                }
            }
            hml_wave_lds_fence();
            // ---------------- out: statistics and candidate maps of the batch's blocks, lane = (chunk, row) again
            if (rel0 >= 0) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int c = k * CPS + sc;
                    const uint32_t fc = f0 + (uint32_t)c;
                    const uint32_t cf = fc * L;
                    const uint32_t off = (uint32_t)(rel0 + sr);   // (rel0 >= 0 here)
                    const uint32_t b = cf + off;
                    if (fc < C && off < B - cf) {
                        const uint32_t* const w = tile + c * PITCH + sr * SLOTW;
                        unsigned long long cm = w[0];
                        if (SLOTW == 4) cm |= (unsigned long long)w[1] << 32;
                        bstat[b] = make_float2(hml_u2f(w[SX]), hml_u2f(w[SX + 1]));
                        hml_tre_store_cand<K>(cand, (uint32_t)b + 1u, cm);
                    }
                }
                hml_wave_lds_fence();
            }
        }
        if (active) {
#pragma unroll
            for (int s = 0; s < K; ++s) exitv[(uint64_t)f * K + s] = alpha[s];
            fb_count[f] = nfb;
            if (nfb) atomicAdd(&mdl->uniform_fallbacks, (unsigned long long)nfb);
            fmap[f] = (unsigned long long)cmap;
        }
    }
}

#endif
