// K10: log_softmax + CTC loss + its gradient w.r.t. the logits, fused (asr.py:148-152:
// F.log_softmax(dim=-1) then nn.CTCLoss(blank=0, reduction='mean', zero_infinity=True)).
//
// The stock PyTorch-ROCm CTC kernels cost ~1.6 ms per step at the benchmark shape (T'=512, B=32, L=100),
// almost all of it latency of a 512-step serial recursion.  Here one workgroup owns one (utterance,
// direction): the alpha (forward) and beta (backward) lattices run concurrently, log-probabilities are
// staged through LDS 64 frames at a time, one __syncthreads per frame.  The gradient
//   dL/dlogit[t][c] = softmax[t][c] - exp(logsumexp_{s: ext[s]=c}(alpha+beta)[t] + nll - logprob[t][c])
// (log_softmax's backward is the identity here because that expression sums to zero over c) is formed by one
// wave per frame with LDS float atomics into the <= 128 class bins.
#include "common.h"
#include <math.h>
#include <type_traits>

#ifndef CTC_SKEW
#define CTC_SKEW 1
#endif
#define CTC_CHUNK 64
#define CTC_MAXV 128
#define NEG_INF (-INFINITY)

// The recursion is a 512-step serial chain, so the two helpers use the hardware exp2 / log2 (v_exp_f32 / v_log_f32,
// ~1 ulp): an absolute error of ~1e-7 per step in the log domain, against per-utterance losses of O(100).
__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    if (m == NEG_INF) return NEG_INF;
    return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(fmaxf(a, b), c);
    if (m == NEG_INF) return NEG_INF;
    return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

// lse[b][t] = logsumexp_c logits[b][t][c]
__global__ void ctc_lse_kernel(const float* __restrict__ logits, float* __restrict__ lse, long rows, int V, int* __restrict__ stall, int B) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < B) stall[r] = 0;               // sticky "a pipeline wait ran into its bound" flags of the lattice kernel (read by grad / mean)
    if (r >= rows) return;
    const float* p = logits + r * V;
    float m = NEG_INF;
    for (int c = 0; c < V; ++c) m = fmaxf(m, p[c]);
    float s = 0.f;
    for (int c = 0; c < V; ++c) s += expf(p[c] - m);
    lse[r] = m + logf(s);
}

// grid (B, 2): y = 0 alpha, y = 1 beta.  lattice[b][t][s] written for t < in_len[b], s < 2*tgt_len[b]+1.
// NS = lattice states per thread: 256 * NS >= 2 * Lmax + 1 (NS = 2 covers transcripts of up to 255 tokens, the
// benchmark's; 4 / 8 / 16 cover up to 511 / 1023 / 2047 -- nn.CTCLoss itself has no limit, asr.py:105).
template <int NS>
__global__ __launch_bounds__(256) void ctc_lattice_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                          const long long* __restrict__ targets, const int* __restrict__ in_len,
                                                          const int* __restrict__ tgt_len, float* __restrict__ alpha,
                                                          float* __restrict__ beta, float* __restrict__ nll, int T, int V, int Lmax,
                                                          int Smax, int blank) {
    extern __shared__ float sm[];
    float* lp = sm;                                  // [CTC_CHUNK][V] log-probs of the current chunk
    float* row = sm + CTC_CHUNK * V;                 // [2][Smax + 4] previous / current lattice row (with guard cells)
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    int Tb = in_len[b];
    if (Tb > T) Tb = T;
    int L = tgt_len[b];
    if (L > Lmax) L = Lmax;
    const int S = 2 * L + 1;
    float* lat = (dir == 0 ? alpha : beta) + (size_t)b * T * Smax;
    const long long* tg = targets + (size_t)b * Lmax;
    const int rs = Smax + 4;
    // guard cells (index 0,1 and S+2,S+3 of the padded row) stay -inf
    for (int i = tid; i < 2 * rs; i += 256) row[i] = NEG_INF;
    __syncthreads();
    if (Tb <= 0) { if (dir == 0 && tid == 0) nll[b] = INFINITY; return; }

    // states handled by this thread: s = tid, tid + 256, ... ; class and skip permission per state
    int cls[NS]; bool skip[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int s = tid + 256 * k;
        cls[k] = blank; skip[k] = false;
        if (s < S && (s & 1)) {
            cls[k] = (int)tg[s >> 1];
            if (dir == 0) skip[k] = s >= 2 && tg[s >> 1] != tg[(s >> 1) - 1];
            else skip[k] = s + 2 < S && tg[s >> 1] != tg[(s >> 1) + 1];
        }
        if (cls[k] < 0 || cls[k] >= V) cls[k] = blank;
    }

    const int nchunks = (Tb + CTC_CHUNK - 1) / CTC_CHUNK;
    for (int ci = 0; ci < nchunks; ++ci) {
        const int c0 = dir == 0 ? ci * CTC_CHUNK : (nchunks - 1 - ci) * CTC_CHUNK;     // first frame of the chunk
        const int cn = min(CTC_CHUNK, Tb - c0);
        __syncthreads();
        for (int i = tid; i < cn * V; i += 256) {
            const int tt = i / V, c = i - tt * V;
            lp[i] = logits[((size_t)b * T + c0 + tt) * V + c] - lse[(size_t)b * T + c0 + tt];
        }
        __syncthreads();
        for (int j = 0; j < cn; ++j) {
            const int tl = dir == 0 ? j : cn - 1 - j;
            const int t = c0 + tl;
            const bool first = dir == 0 ? (t == 0) : (t == Tb - 1);
            const int cur = t & 1, prv = cur ^ 1;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int s = tid + 256 * k;
                if (s < S) {
                    const float e = lp[tl * V + cls[k]];
                    float v;
                    if (first) {
                        const bool start = dir == 0 ? (s <= 1) : (s >= S - 2);
                        v = start ? e : NEG_INF;
                    } else if (dir == 0) {
                        const float* pr = row + prv * rs + 2 + s;
                        v = e + lse3(pr[0], pr[-1], skip[k] ? pr[-2] : NEG_INF);
                    } else {
                        const float* pr = row + prv * rs + 2 + s;
                        v = e + lse3(pr[0], pr[1], skip[k] ? pr[2] : NEG_INF);
                    }
                    row[cur * rs + 2 + s] = v;
                    lat[(size_t)t * Smax + s] = v;
                }
            }
            __syncthreads();
        }
    }
    if (dir == 0 && tid == 0) {
        const float* last = row + ((Tb - 1) & 1) * rs + 2;
        const float ll = S >= 2 ? lse2(last[S - 1], last[S - 2]) : last[S - 1];
        nll[b] = -ll;
    }
}

// ---------------------------------------------------------------------------------------------
// The same lattice as a WAVE PIPELINE (2 * Lmax + 1 <= 1024 states).  The kernel above pays, on every one of the T dependent
// frames, an LDS write -> barrier -> LDS read round trip of the whole row across 4 waves (~0.33 us per frame: 170 us at
// T' = 512).  Here a lane owns ONE state in direction order (u = s for alpha, u = S-1-s for beta, so both recursions read
// u, u-1, u-2), keeps it in a register and gets its neighbours by DPP (wave_shr:1); only the two states at a wave's upper edge
// go through LDS, into a ring slot per frame, and the waves run SKEWED: wave w may start frame q as soon as wave w-1 has
// published frame q-1 (a progress word per wave; no workgroup barrier in the recursion).  Emission log-probabilities are
// gathered straight from global memory eight frames ahead.
// (Round 5, measured at B = 32, T' = 512, 100-token targets, whole head 134 us: timing-only ablations (CTC_ABL) put 62 us on the
//  handshake, 37 us on the four transcendentals per frame, 10-12 us each on the lattice store and the emission gathers; the same
//  pipeline in LOCKSTEP -- fixed skew of two frames, one s_barrier per frame, no progress words -- was built and measured SLOWER,
//  143 us: a four-wave s_barrier costs more per frame (~75 ns) than the polling it replaces.  DESIGN.md section 8.)
// log2-domain log-sum-exp of three terms on the raw v_exp_f32 / v_log_f32 (no range scaling: the sum lies in [1, 3] or is exactly
// 0), branch-free: an all -inf input gives mm + log2(0) = -inf, never inf - inf
#ifndef CTC_ABL
#define CTC_ABL 0      /* timing-only ablations (wrong results): 1 no lattice store, 2 no transcendentals, 4 no handshake, 8 no emission loads */
#endif
__device__ __forceinline__ float ctc_lse3_log2(float a, float b, float c) {
    if constexpr (CTC_ABL & 2) return (a + b + c) * 0.3f;
    const float mm = fmaxf(fmaxf(fmaxf(a, b), c), -1e30f);
    const float sum = __builtin_amdgcn_exp2f(a - mm) + __builtin_amdgcn_exp2f(b - mm) + __builtin_amdgcn_exp2f(c - mm);
    return mm + __builtin_amdgcn_logf(sum);
}
#define CTC_RING 32
#define CTC_SPIN_MAX (1 << 20)      /* a wave never waits forever on a neighbour (all waves of a workgroup are resident: this bound is never reached) */
#define CTC_PF 8
__device__ __forceinline__ float ctc_wave_shr1(float fill, float v) {       // lane l gets v of lane l-1; lane 0 gets `fill`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}

__global__ __launch_bounds__(1024) void ctc_lattice_skew_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                                const long long* __restrict__ targets, const int* __restrict__ in_len,
                                                                const int* __restrict__ tgt_len, float* __restrict__ alpha,
                                                                float* __restrict__ beta, float* __restrict__ nll, int T, int V, int Lmax,
                                                                int Smax, int blank, int* __restrict__ stall, const int* __restrict__ only) {
    __shared__ float bnd[16][CTC_RING][2];
    __shared__ int progress[16];
    __shared__ float fin[2];
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    if (only && !only[b]) return;            // fallback launch behind the linear-domain kernels: only the utterances they flagged
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, NW = blockDim.x >> 6;
    int Tb = in_len[b];
    if (Tb > T) Tb = T;
    int L = tgt_len[b];
    if (L > Lmax) L = Lmax;
    const int S = 2 * L + 1;
    float* lat = (dir == 0 ? alpha : beta) + (size_t)b * T * Smax;
    const long long* tg = targets + (size_t)b * Lmax;
    if (lane == 0) progress[w] = -1;
    if (tid < 2) fin[tid] = NEG_INF;
    __syncthreads();
    if (Tb <= 0) { if (dir == 0 && tid == 0) nll[b] = INFINITY; return; }

    const int u = tid;                                   // state in direction order
    const int sidx = dir == 0 ? u : S - 1 - u;           // lattice state
    const bool active = u < S;
    int cls = blank;
    bool skip = false;
    if (active && (sidx & 1)) {
        cls = (int)tg[sidx >> 1];
        if (dir == 0) skip = sidx >= 2 && tg[sidx >> 1] != tg[(sidx >> 1) - 1];
        else skip = sidx + 2 < S && tg[sidx >> 1] != tg[(sidx >> 1) + 1];
    }
    if (cls < 0 || cls >= V) cls = blank;

    // frame q of the recursion is time t = q (alpha) / Tb-1-q (beta); emissions of frames q0+8 .. q0+15 are in flight while q0 .. q0+7 compute
    const float* lg = logits + (size_t)b * T * V + cls;
    const float* ls = lse + (size_t)b * T;
    auto emission = [&](int q) -> float {
        const int qq = q < Tb ? q : Tb - 1;
        const int t = dir == 0 ? qq : Tb - 1 - qq;
        if constexpr (CTC_ABL & 8) return -3.f - 0.001f * (float)t;
        return (lg[(size_t)t * V] - ls[t]) * 1.44269504088896340736f;       // log2 units: the recursion runs on exp2 / log2 directly
    };
    float cur[CTC_PF], nxt[CTC_PF];
#pragma unroll
    for (int j = 0; j < CTC_PF; ++j) cur[j] = emission(j);

    // relaxed workgroup-scope atomics: plain ds_read / ds_write that the compiler neither caches nor fences (a `volatile` access
    // is bracketed by vmcnt(0) waits, i.e. by the acknowledgement of the previous frame's lattice store: 240 us instead of 170).
    // LDS executes one wave's operations in order, so "edge values, then progress word" needs no fence -- only program order.
    auto prog_load = [&](int i) { return __hip_atomic_load(&progress[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    float a_prev = NEG_INF;
    int pnext = -2;
    float n63 = NEG_INF, n62 = NEG_INF;
    // lattice row pointer of frame q, advanced by one row per frame (alpha walks up, beta down)
    float* lrow = lat + (size_t)(dir == 0 ? 0 : Tb - 1) * Smax + sidx;
    const long lstep = dir == 0 ? (long)Smax : -(long)Smax;
    // one frame; UP: there is an upstream wave (w > 0), DOWN: a downstream wave (w < NW - 1), FIRST: q == 0.  The roles are fixed
    // per wave, so the loop below is instantiated per role instead of testing them on every frame.
    auto frame = [&](auto up_t, auto down_t, auto first_t, int q, float e) {
        constexpr bool UP = decltype(up_t)::value, DOWN = decltype(down_t)::value, FIRST = decltype(first_t)::value;
        float v63 = NEG_INF, v62 = NEG_INF;
        if constexpr (UP) {
            if constexpr (!FIRST) {
                // (the progress word travels as an opaque VGPR until here: as a plain uniform value hipcc moves it to an SGPR --
                //  and waits for the LDS read -- right where it was issued, which puts the read's latency back on the chain)
                asm volatile("" : "+v"(pnext));
                if (__builtin_amdgcn_readfirstlane(pnext) >= q - 1) {    // the copy fetched during the previous frame is valid
                    v63 = n63; v62 = n62;
                } else {
                    // wait until the upstream wave is TWO frames ahead (or done): with a skew of one, the speculative fetch
                    // below would find frame q unpublished on every frame and this slow path would run each time
                    const int need = min(q + 1, Tb - 1);
                    int guard = 0;
                    for (; prog_load(w - 1) < need && guard < CTC_SPIN_MAX; ++guard) __builtin_amdgcn_s_sleep(1);
                    if (guard >= CTC_SPIN_MAX && lane == 0) stall[b] = 1;      // never reached by design; if it is, poison the result (no silent wrong loss)
                    asm volatile("" ::: "memory");
                    v63 = bnd[w - 1][(q - 1) & (CTC_RING - 1)][0];
                    v62 = bnd[w - 1][(q - 1) & (CTC_RING - 1)][1];
                }
            }
            // speculative fetch for frame q + 1 (edge values of frame q): progress word FIRST -- if it already says >= q, the slot
            // read after it (LDS keeps a wave's operations in order) is complete; used next frame, so its latency hides here
            pnext = prog_load(w - 1);
            asm volatile("" ::: "memory");
            n63 = bnd[w - 1][q & (CTC_RING - 1)][0];
            n62 = bnd[w - 1][q & (CTC_RING - 1)][1];
        }
        float v;
        if constexpr (FIRST) {
            v = (u <= 1) ? e : NEG_INF;
        } else {
            const float sm1 = ctc_wave_shr1(v63, a_prev);
            const float sm2 = ctc_wave_shr1(v62, sm1);
            v = e + ctc_lse3_log2(a_prev, sm1, skip ? sm2 : NEG_INF);
        }
        if (!active) v = NEG_INF;
        if constexpr (DOWN) {
            // back-pressure, once per half ring: frames q .. q + RING/2 - 1 reuse the slots of frames q - RING .. q - RING/2 - 1, which
            // wave w+1 has read once it has published frame q - RING/2
            if ((q & (CTC_RING / 2 - 1)) == 0 && q >= CTC_RING / 2) {
                int guard = 0;
                for (; prog_load(w + 1) < q - CTC_RING / 2 && guard < CTC_SPIN_MAX; ++guard) __builtin_amdgcn_s_sleep(1);
                if (guard >= CTC_SPIN_MAX && lane == 0) stall[b] = 1;
            }
            asm volatile("" ::: "memory");
            if (lane >= 62) bnd[w][q & (CTC_RING - 1)][63 - lane] = v;
        }
        asm volatile("" ::: "memory");       // program order only: LDS performs one wave's writes in the order they were issued
        if (lane == 0) __hip_atomic_store(&progress[w], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        a_prev = v;
        if (!(CTC_ABL & 1)) if (active) *lrow = v * 0.69314718055994530942f;      // the lattice is stored in natural-log units
        lrow += lstep;
    };
    auto run = [&](auto up_t, auto down_t) {
        for (int q0 = 0; q0 < Tb; q0 += CTC_PF) {
#pragma unroll
            for (int j = 0; j < CTC_PF; ++j) nxt[j] = emission(q0 + CTC_PF + j);
            if (q0 == 0) {
                frame(up_t, down_t, std::true_type{}, 0, cur[0]);
#pragma unroll
                for (int j = 1; j < CTC_PF; ++j)
                    if (j < Tb) frame(up_t, down_t, std::false_type{}, j, cur[j]);
            } else if (q0 + CTC_PF <= Tb) {
#pragma unroll
                for (int j = 0; j < CTC_PF; ++j) frame(up_t, down_t, std::false_type{}, q0 + j, cur[j]);
            } else {
#pragma unroll
                for (int j = 0; j < CTC_PF; ++j)
                    if (q0 + j < Tb) frame(up_t, down_t, std::false_type{}, q0 + j, cur[j]);
            }
#pragma unroll
            for (int j = 0; j < CTC_PF; ++j) cur[j] = nxt[j];
        }
    };
    const bool has_up = !(CTC_ABL & 4) && w > 0, has_down = !(CTC_ABL & 4) && w < NW - 1;
    if (has_up && has_down) run(std::true_type{}, std::true_type{});
    else if (has_up) run(std::true_type{}, std::false_type{});
    else if (has_down) run(std::false_type{}, std::true_type{});
    else run(std::false_type{}, std::false_type{});
    if (dir == 0) {
        if (u == S - 1) fin[0] = a_prev;
        if (S >= 2 && u == S - 2) fin[1] = a_prev;
        __syncthreads();
        if (tid == 0) nll[b] = -(S >= 2 ? lse2(fin[0] * 0.69314718055994530942f, fin[1] * 0.69314718055994530942f) : fin[0] * 0.69314718055994530942f);
    }
}

// one wave per (b, t): grad[b][t][c] = softmax - exp(lcab + nll - logprob); zero for padded frames / infeasible utterances
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                       const long long* __restrict__ targets, const int* __restrict__ in_len,
                                                       const int* __restrict__ tgt_len, const float* __restrict__ alpha,
                                                       const float* __restrict__ beta, float* nll,
                                                       float* __restrict__ grad, int B, int T, int V, int Lmax, int Smax, int blank,
                                                       int mean_scale, const int* __restrict__ stall, float* __restrict__ loss,
                                                       const int* __restrict__ only) {
    __shared__ float bins[4][CTC_MAXV];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + wave;
    if (loss && blockIdx.x == 0 && wave == 0) {
        // loss = mean_b( finite(nll_b) ? nll_b / max(len_b, 1) : 0 )   (reduction='mean', zero_infinity=True: only +-inf is zeroed,
        // a NaN -- or a lattice whose pipeline wait hit its bound -- stays visible)
        float s = 0.f;
        for (int bb = lane; bb < B; bb += 64) {
            const float v = stall[bb] ? __builtin_nanf("") : nll[bb];
            if (!(v == INFINITY || v == -INFINITY)) s += v / (float)max(tgt_len[bb], 1);
        }
        s = wave_sum(s);
        if (lane == 0) loss[0] = s / (float)B;
    }
    // `only` (the fallback launch behind the linear-domain pair): a SMALL grid -- one wave per frame index, looping over the flagged
    // utterances (normally none: 128 workgroups that read B flags and leave, instead of B T / 4 workgroups that each leave)
    if (only) {
        if (w >= T) return;
    } else if (w >= (long)B * T) return;
    auto frame = [&](const int b, const int t) {
    float* g = grad + ((size_t)b * T + t) * V;
    if (stall[b]) {                                   // see ctc_lattice_skew_kernel: poison instead of a silently wrong result
        for (int c = lane; c < V; c += 64) g[c] = __builtin_nanf("");
        if (t == 0 && lane == 0) nll[b] = __builtin_nanf("");
        return;
    }
    const float n = nll[b];
    int Tb = in_len[b];
    if (Tb > T) Tb = T;
    if (t >= Tb || !(n < INFINITY)) {                 // padded frame, or zero_infinity
        for (int c = lane; c < V; c += 64) g[c] = 0.f;
        return;
    }
    int L = tgt_len[b];
    if (L > Lmax) L = Lmax;
    const int S = 2 * L + 1;
    const float* al = alpha + ((size_t)b * T + t) * Smax;
    const float* be = beta + ((size_t)b * T + t) * Smax;
    const long long* tg = targets + (size_t)b * Lmax;
    for (int c = lane; c < V; c += 64) bins[wave][c] = 0.f;
    float m = NEG_INF;
    for (int s = lane; s < S; s += 64) m = fmaxf(m, al[s] + be[s]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");       // wave-local LDS hand-off between lanes: keep program order for the compiler too
    if (m > NEG_INF) {
        for (int s = lane; s < S; s += 64) {
            int c = (s & 1) ? (int)tg[s >> 1] : blank;
            if (c < 0 || c >= V) c = blank;
            atomicAdd(&bins[wave][c], expf(al[s] + be[s] - m));
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");       // wave-local LDS hand-off between lanes: keep program order for the compiler too
    const float z = lse[(size_t)b * T + t];
    const float* lg = logits + ((size_t)b * T + t) * V;
    // mean_scale: gradient of CTCLoss(reduction='mean') = mean_b(nll_b / max(len_b, 1)) instead of nll_b
    const float gs = mean_scale ? 1.f / ((float)max(tgt_len[b], 1) * (float)B) : 1.f;
    for (int c = lane; c < V; c += 64) {
        const float lpc = lg[c] - z;
        const float acc = bins[wave][c];
        const float occ = (acc > 0.f) ? expf(m + logf(acc) + n - lpc) : 0.f;
        g[c] = (expf(lpc) - occ) * gs;
    }
    };
    if (only) {
        for (int b = 0; b < B; ++b)
            if (only[b]) {
                frame(b, (int)w);
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");       // bins[wave] is reused by the next flagged utterance
            }
    } else {
        frame((int)(w / T), (int)(w % T));
    }
}


// ---------------------------------------------------------------------------------------------
// Round 6: the lattices in the LINEAR domain, fp64, one wave per (utterance, direction), four states per lane.
//
// Why.  The wave pipeline above spends 106 us on T' = 512 frames: ~210 cycles of dependent log-sum-exp arithmetic per frame (three
// v_exp_f32 + one v_log_f32 on the chain) plus a cross-wave hand-over per frame (62 of the head's 134 us, round 5).  In the linear
// domain a frame is a_t(u) = (a(u) + a(u-1) + skip(u) a(u-2)) p_t(class(u)): two adds and a multiply.  Round 2 tried that in fp32
// with one power-of-two scale per frame and lost paths that sit more than 2^-126 below the frame's maximum but still carry the
// final probability.  fp64 moves that limit to 2^-1022, and a CHECK makes the result exact regardless: the gradient kernel forms,
// for every frame, sum_s alpha_t(s) beta_t(s) / p_t(class(s)) ... which must equal P for every t (each path passes exactly one
// state per frame); if any frame of an utterance disagrees with the utterance's P by more than 1e-5 in log2 units -- mass was
// flushed somewhere, in either direction -- the utterance is flagged and the two log-domain kernels above run for it behind this
// pair (they run for nobody otherwise: two near-empty launches).  256 states (transcripts of up to 127 tokens) fit one wave at four
// per lane: the neighbours u-1 / u-2 are registers of the same lane, only a lane's first two states need lane-1's last two (two
// 64-bit DPP moves), there is no hand-over at all.  Emission probabilities are staged per chunk of frames into LDS as doubles
// (mantissa by v_exp_f32 of the fractional part of log2 p, exponent by ldexp: exact to fp32's exp2, any magnitude); the common scale
// is renewed every 16 frames from the exponent of the wave's largest state (a 32-bit DPP max over the high dwords) and its
// cumulative exponent stored per frame beside the lattice row.
// Cost per frame: ~16 fp64 VALU + 4 DPP + 4 ds_read_b64 + 2 stores for the four states = ~130 issue cycles for a lone wave.
#define CTCL_MAXS 256
#define CTCL_ROW (CTCL_MAXS + 64)      /* 4-byte words per lattice row: 256 state floats + 64 lane exponents */
#define CTCL_RESCALE 4
#ifndef CTC_LIN
#define CTC_LIN 1
#endif
#ifndef CTCL_ABL
#define CTCL_ABL 0      /* timing-only ablations of the linear lattice (wrong results): 1 no lattice stores, 2 no rescale, 4 no staging */
#endif
__device__ __forceinline__ double ctcl_shr1(double v) {       // lane l gets v of lane l-1; lane 0 gets 0.0
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// largest high dword over the wave (non-negative doubles order like their high dwords) on the DPP path -- six VALU max steps, no LDS
// crossbar round trips (a __shfl_xor is a ds_bpermute) -- returned wave-uniform
__device__ __forceinline__ unsigned ctcl_wave_max_u32(unsigned v) {
#define CTCL_MAX_STEP(CTRL, ROWS) v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, true))
    CTCL_MAX_STEP(0xB1, 0xf);       // quad_perm [1,0,3,2]
    CTCL_MAX_STEP(0x4E, 0xf);       // quad_perm [2,3,0,1]
    CTCL_MAX_STEP(0x141, 0xf);      // row_half_mirror
    CTCL_MAX_STEP(0x140, 0xf);      // row_mirror
    CTCL_MAX_STEP(0x142, 0xa);      // row_bcast:15 into rows 1, 3
    CTCL_MAX_STEP(0x143, 0xc);      // row_bcast:31 into rows 2, 3
#undef CTCL_MAX_STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// lat [2][B][T][CTCL_MAXS] doubles in DIRECTION order (alpha: u = s; beta: u = S-1-s), ecum [2][B][T] ints: true value = lat * 2^-ecum
// Four waves per (utterance, direction): wave 0 runs the recursion of chunk k out of one LDS buffer while waves 1 - 3 stage chunk
// k + 1 into the other (all four stage chunk 0): with ONE wave doing both, the staging -- 116 elements per lane and chunk behind
// their global-load latency -- was 31 of the kernel's 99 us (timing-only ablations, CTCL_ABL).  One workgroup barrier per chunk.
__global__ __launch_bounds__(256) void ctc_lattice_lin_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                              const long long* __restrict__ targets, const int* __restrict__ in_len,
                                                              const int* __restrict__ tgt_len, float* __restrict__ lat,
                                                              float* __restrict__ nll, double* __restrict__ ll2d,
                                                              int* __restrict__ bad, int B, int T, int V, int Lmax, int blank, int chunk) {
    extern __shared__ double pe_all[];               // [2][chunk + 1][V + 1]: p_t(c); column V is 0.0 (inactive states)
    __shared__ double fin[CTCL_MAXS];
    __shared__ int fine[64];
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int Tb = in_len[b];
    if (Tb > T) Tb = T;
    int L = tgt_len[b];
    if (L > Lmax) L = Lmax;
    const int S = 2 * L + 1;
    if (dir == 0 && tid == 0) bad[b] = 0;
    if (Tb <= 0) { if (dir == 0 && tid == 0) { nll[b] = INFINITY; ll2d[b] = -INFINITY; } return; }
    const int VP = V + 1;
    const size_t bufn = (size_t)(chunk + 1) * VP;
    // ---- staging of frames [c0, c0 + cn) of this direction into `pe` by threads `id` of `nth` (coalesced: the chunk is a contiguous
    //      run of cn * V logits, ascending t for alpha, descending for beta; eight loads per thread in flight; the spare row keeps
    //      whatever it holds except for its zero column -- it is read into registers and never used) ----
    auto stage = [&](double* pe, int c0, int id, int nth) {
        const int cn = min(chunk, Tb - c0);
        if (cn <= 0) return;
        const int t_lo = dir == 0 ? c0 : Tb - c0 - cn;
        const float* lg0 = logits + ((size_t)b * T + t_lo) * V;
        const float* ls0 = lse + (size_t)b * T + t_lo;
        const int n = cn * V;
        const float rv = 1.f / (float)V;
        for (int base = 0; base < ((CTCL_ABL & 4) ? 8 * nth : n); base += 8 * nth) {
            float lv[8], zv[8];
            int tl[8], cc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ii = min(base + id + nth * u, n - 1);
                int q_ = (int)(((float)ii + 0.5f) * rv);             // ii / V without the integer division (exact after the fix-up)
                int c_ = ii - q_ * V;
                if (c_ < 0) { c_ += V; --q_; } else if (c_ >= V) { c_ -= V; ++q_; }
                tl[u] = q_; cc[u] = c_;
                lv[u] = lg0[ii];
                zv[u] = ls0[q_];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float lp2 = (lv[u] - zv[u]) * 1.44269504088896340736f;
                const float fl = floorf(lp2);
                double p = 0.0;
                if (lp2 > -3000.f) p = ldexp((double)__builtin_amdgcn_exp2f(lp2 - fl), (int)fl);
                else if (lp2 != lp2) p = (double)lp2;                          // NaN logits stay NaN (the check then flags the utterance)
                const int tt = dir == 0 ? tl[u] : cn - 1 - tl[u];
                if (base + id + nth * u < n) pe[tt * VP + cc[u]] = p;
            }
        }
        for (int tt = id; tt <= cn; tt += nth) pe[tt * VP + V] = 0.0;          // the zero column (inactive states), spare row included
    };
    stage(pe_all, 0, tid, 256);
    __syncthreads();

    const long long* tg = targets + (size_t)b * Lmax;
    int cidx[4];
    double skipm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int u = 4 * lane + j;
        const int sidx = dir == 0 ? u : S - 1 - u;
        int cls = blank;
        bool skip = false;
        if (u < S && (sidx & 1)) {
            cls = (int)tg[sidx >> 1];
            if (dir == 0) skip = sidx >= 2 && tg[sidx >> 1] != tg[(sidx >> 1) - 1];
            else skip = sidx + 2 < S && tg[sidx >> 1] != tg[(sidx >> 1) + 1];
        }
        if (cls < 0 || cls >= V) cls = blank;
        cidx[j] = u < S ? cls : V;
        skipm[j] = skip ? 1.0 : 0.0;
    }
    // Lattice row of frame q: CTCL_ROW 4-byte words -- 256 floats (the lane's four states relative to ITS exponent) + 64 ints (the
    // lanes' exponents): true value of state 4 l + j = row[4 l + j] * 2^-rowe[l].  One 16-byte and one 4-byte store per lane and frame.
    float* lptr = lat + ((size_t)(dir * B + b) * T + (dir == 0 ? 0 : Tb - 1)) * CTCL_ROW;
    const long lstep = dir == 0 ? (long)CTCL_ROW : -(long)CTCL_ROW;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int ecl = 0;                                       // this lane's exponent: true = a * 2^-ecl
    auto store_row = [&]() {
        if (!(CTCL_ABL & 1)) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<f4*>(lptr + 4 * lane) = f4{(float)a0, (float)a1, (float)a2, (float)a3};
            reinterpret_cast<int*>(lptr)[CTCL_MAXS + lane] = ecl;
        }
        lptr += lstep;
    };
    // Per-LANE renormalisation (every CTCL_RESCALE frames): bring the lane's largest state to [1, 2).  One scale per wave is not
    // enough for real networks: a blank-collapsed model (every CTC run passes through that phase) has the all-blank states 2^-13 per
    // label above the states that carry the alignment -- 2^-1300 at 100 tokens, beyond fp64 -- while four ADJACENT states never
    // differ by more than a few label probabilities.  Lanes the lattice's front has not reached yet (all zero) take the exponent of
    // the front lane, so that the front's values arrive in range.
    auto rescale = [&]() {
        const unsigned h = max(max((unsigned)(__builtin_bit_cast(unsigned long long, a0) >> 32), (unsigned)(__builtin_bit_cast(unsigned long long, a1) >> 32)),
                               max((unsigned)(__builtin_bit_cast(unsigned long long, a2) >> 32), (unsigned)(__builtin_bit_cast(unsigned long long, a3) >> 32)));
        const int ex = (int)((h >> 20) & 0x7ff);               // biased exponent of the lane's largest state (0: all zero / denormal)
        const bool live = ex != 0;
        const int k = (live && ex != 0x7ff) ? 1023 - ex : 0;
        a0 = ldexp(a0, k); a1 = ldexp(a1, k); a2 = ldexp(a2, k); a3 = ldexp(a3, k);
        ecl += k;
        const unsigned long long m = __ballot(live);
        if (m != 0ull) {                                       // (wave-uniform)
            const int front = 63 - __builtin_clzll(m);
            const int ecf = __builtin_amdgcn_readlane(ecl, front);
            if (!live) ecl = ecf;
        }
    };
    // lane l-1's last two states, brought to THIS lane's exponent (lane 0 has no predecessor: zeros)
    auto incoming = [&](double& p3, double& p2) {
        const int ecp = __builtin_amdgcn_update_dpp(ecl, ecl, 0x138, 0xf, 0xf, false);      // lane l-1's exponent (lane 0: its own)
        const int sh = min(ecl - ecp, 900);                    // (a clamped shift under-scales: the mass check then flags the utterance)
        p3 = ldexp(ctcl_shr1(a3), sh);
        p2 = ldexp(ctcl_shr1(a2), sh);
    };
    int kbuf = 0;
    for (int c0 = 0; c0 < Tb; c0 += chunk, kbuf ^= 1) {
        const int cn = min(chunk, Tb - c0);
        if (wave != 0) {
            stage(pe_all + (kbuf ^ 1) * bufn, c0 + chunk, tid - 64, 192);     // the NEXT chunk, beside the recursion
        } else {
            const double* row = pe_all + kbuf * bufn;
            double e0 = row[cidx[0]], e1 = row[cidx[1]], e2 = row[cidx[2]], e3 = row[cidx[3]];
            int tt = 0;
            if (c0 == 0) {                                     // frame 0: states 0 and 1 start the lattice
                a0 = lane == 0 ? e0 : 0.0;
                a1 = lane == 0 ? e1 : 0.0;
                a2 = 0.0; a3 = 0.0;
                store_row();
                row += VP;
                e0 = row[cidx[0]]; e1 = row[cidx[1]]; e2 = row[cidx[2]]; e3 = row[cidx[3]];
                tt = 1;
            }
            for (; tt < cn; ++tt) {
                row += VP;
                const double f0 = row[cidx[0]], f1 = row[cidx[1]], f2 = row[cidx[2]], f3 = row[cidx[3]];      // the NEXT frame's (used a frame later)
                double p3, p2;
                incoming(p3, p2);
                const double n0 = fma(skipm[0], p2, a0 + p3);
                const double n1 = fma(skipm[1], p3, a1 + a0);
                const double n2 = fma(skipm[2], a0, a2 + a1);
                const double n3 = fma(skipm[3], a1, a3 + a2);
                a0 = n0 * e0; a1 = n1 * e1; a2 = n2 * e2; a3 = n3 * e3;
                if (!(CTCL_ABL & 2) && ((c0 + tt) & (CTCL_RESCALE - 1)) == CTCL_RESCALE - 1) rescale();
                store_row();
                e0 = f0; e1 = f1; e2 = f2; e3 = f3;
            }
        }
        __syncthreads();
    }
    if (dir == 0 && wave == 0) {
        fin[4 * lane + 0] = a0; fin[4 * lane + 1] = a1; fin[4 * lane + 2] = a2; fin[4 * lane + 3] = a3;
        fine[lane] = ecl;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) {
            const int e1 = fine[(S - 1) >> 2], e2 = S >= 2 ? fine[(S - 2) >> 2] : e1;
            const int E = min(e1, e2);
            const double tot = ldexp(fin[S - 1], E - e1) + (S >= 2 ? ldexp(fin[S - 2], E - e2) : 0.0);
            // log2 P = log2(tot) - E ; tot == 0 -> +inf nll (infeasible, or flushed: the gradient kernel's check then flags it)
            const double l2 = tot > 0.0 ? log2(tot) - (double)E : -INFINITY;           // kept in double for the gradient kernel's check
            ll2d[b] = l2;
            nll[b] = tot > 0.0 ? (float)(-l2 * 0.69314718055994530942) : INFINITY;
        }
    }
}

// one wave per (b, t): gradient from the linear lattices, and the per-frame mass check (see above)
__global__ __launch_bounds__(256) void ctc_grad_lin_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                           const long long* __restrict__ targets, const int* __restrict__ in_len,
                                                           const int* __restrict__ tgt_len, const float* __restrict__ lat,
                                                           const float* __restrict__ nll,
                                                           const double* __restrict__ ll2d, float* __restrict__ grad, int* __restrict__ bad,
                                                           int B, int T, int V, int Lmax, int blank, int mean_scale) {
    __shared__ double bins[4][CTC_MAXV];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + wave;
    if (w >= (long)B * T) return;
    const int b = (int)(w / T), t = (int)(w % T);
    float* g = grad + ((size_t)b * T + t) * V;
    const float n = nll[b];
    int Tb = in_len[b];
    if (Tb > T) Tb = T;
    if (t >= Tb) {                                    // padded frame
        for (int c = lane; c < V; c += 64) g[c] = 0.f;
        return;
    }
    int L = tgt_len[b];
    if (L > Lmax) L = Lmax;
    const int S = 2 * L + 1;
    const float* al = lat + ((size_t)(0 * B + b) * T + t) * CTCL_ROW;
    const float* be = lat + ((size_t)(1 * B + b) * T + t) * CTCL_ROW;
    const int* ale = reinterpret_cast<const int*>(al) + CTCL_MAXS;
    const int* bee = reinterpret_cast<const int*>(be) + CTCL_MAXS;
    const long long* tg = targets + (size_t)b * Lmax;
    for (int c = lane; c < V; c += 64) bins[wave][c] = 0.0;
    // alpha_t(s) beta_t(s) = prod 2^-e per state, each with its own exponent (the two lanes' exponents): first the smallest exponent
    // (largest scale) over the states with a non-zero product, then the per-class sums relative to it (down-shifts only)
    double prod[4];
    int pe_[4], pc_[4];
    int emin = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = lane + 64 * i;
        prod[i] = 0.0; pe_[i] = 0; pc_[i] = blank;
        if (s < S) {
            const int ub = S - 1 - s;
            prod[i] = (double)al[s] * (double)be[ub];
            pe_[i] = ale[s >> 2] + bee[ub >> 2];
            int c = (s & 1) ? (int)tg[s >> 1] : blank;
            if (c < 0 || c >= V) c = blank;
            pc_[i] = c;
            if (prod[i] > 0.0) emin = min(emin, pe_[i]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) emin = min(emin, __shfl_xor(emin, off, 64));
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (prod[i] != 0.0 || prod[i] != prod[i]) atomicAdd(&bins[wave][pc_[i]], ldexp(prod[i], max(emin - pe_[i], -2000)));
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const float z = lse[(size_t)b * T + t];
    const float* lg = logits + ((size_t)b * T + t) * V;
    const int esum = emin == 0x7fffffff ? 0 : emin;
    const double ll2 = ll2d[b];                        // log2 P from the alpha lattice, in double
    // sum over classes of (sum_{s in c} alpha beta) / p_t(c) = P for every frame: the check, relative to P
    double tot = 0.0;
    float occ[2] = {0.f, 0.f};                         // alpha beta / (P p_t(c)) for this lane's (up to two) classes
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = lane + 64 * i;
        if (c < V) {
            const double acc = bins[wave][c];
            const float lp2 = (lg[c] - z) * 1.44269504088896340736f;          // the value the lattice kernel staged
            if (acc > 0.0 && ll2 > -INFINITY) {
                int ex;
                const double mant = frexp(acc, &ex);                            // acc = mant 2^ex, mant in [0.5, 1)
                const double l2 = (double)(ex - esum) - (double)lp2 - ll2;        // log2 of the share, without the mantissa: O(1) for real shares
                const double share = l2 > -1000.0 ? mant * exp2(l2) : 0.0;
                tot += share;
                occ[i] = (float)share;
            } else if (acc != acc) {
                tot = acc;                                                      // NaN propagates into the check
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
    const bool feasible = n < INFINITY;
    // a frame whose mass is not P (the states are stored as floats: within 1e-5 relative), or an utterance whose alpha came out empty
    // although it is not OBVIOUSLY infeasible, goes to the exact log-domain kernels (NaN fails the comparison and is flagged too)
    const bool ok = feasible ? (fabs(tot - 1.0) <= 1e-5) : (Tb < L);
    if (!ok && lane == 0) bad[b] = 1;
    const float gs = mean_scale ? 1.f / ((float)max(tgt_len[b], 1) * (float)B) : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = lane + 64 * i;
        if (c < V) {
            const float lpc = lg[c] - z;
            g[c] = feasible ? (expf(lpc) - occ[i]) * gs : 0.f;
        }
    }
}

static bool ctc_lin_enabled() {
    static const bool on = [] { const char* e = getenv("V100_CTC_LIN"); return CTC_LIN && !(e && e[0] == '0'); }();
    return on;
}
// floats of the log-domain part (alpha + beta + lse + stall flags), and of the linear-domain part behind it (lattices as doubles,
// log2 P per utterance as a double, cumulative exponents and the "flagged" words as ints; 2 floats of slack for the 8-byte alignment)
static long ctc_ws_log(int B, int T, int Lmax) { return 2L * B * T * (2 * Lmax + 1) + (long)B * T + B; }
static long ctc_ws_lin(int B, int T) { return 2L * B * T * CTCL_ROW + 2L * B + B + 8; }
extern "C" int v100_ctc_workspace_floats(int B, int T, int Lmax) {
    long n = ctc_ws_log(B, T, Lmax);
    if (ctc_lin_enabled() && 2 * Lmax + 1 <= CTCL_MAXS) n += ctc_ws_lin(B, T);
    return n > 0x7fffffffL ? -1 : (int)n;
}

static int ctc_run(const float* logits, const long long* targets, const int* in_len, const int* tgt_len, float* workspace, float* nll,
                   float* loss, float* grad, int B, int T, int V, int Lmax, int blank, void* stream);

extern "C" int v100_ctc_loss(const float* logits, const long long* targets, const int* in_len, const int* tgt_len, float* workspace,
                             float* nll, float* grad, int B, int T, int V, int Lmax, int blank, void* stream) {
    return ctc_run(logits, targets, in_len, tgt_len, workspace, nll, nullptr, grad, B, T, V, Lmax, blank, stream);
}

// The same plus the reduction of nn.CTCLoss(reduction='mean', zero_infinity=True): loss[0] = mean_b(nll_b / max(len_b, 1)) over
// the feasible utterances, grad = d loss / d logits (asr.py:105, 152) -- no host-side elementwise kernels around the call.
extern "C" int v100_ctc_loss_mean(const float* logits, const long long* targets, const int* in_len, const int* tgt_len, float* workspace,
                                  float* nll, float* loss, float* grad, int B, int T, int V, int Lmax, int blank, void* stream) {
    if (!loss) return V100_ERR_NULL;
    return ctc_run(logits, targets, in_len, tgt_len, workspace, nll, loss, grad, B, T, V, Lmax, blank, stream);
}

static int ctc_run(const float* logits, const long long* targets, const int* in_len, const int* tgt_len, float* workspace, float* nll,
                   float* loss, float* grad, int B, int T, int V, int Lmax, int blank, void* stream) {
    if (!logits || !targets || !in_len || !tgt_len || !workspace || !nll || !grad) return V100_ERR_NULL;
    if (B <= 0 || T <= 0 || V <= 0 || V > CTC_MAXV || Lmax < 0 || 2 * Lmax + 1 > 4096 || blank < 0 || blank >= V) return V100_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int Smax = 2 * Lmax + 1;
    float* alpha = workspace;
    float* beta = alpha + (size_t)B * T * Smax;
    float* lse = beta + (size_t)B * T * Smax;
    const long rows = (long)B * T;
    int* stall = (int*)(lse + (size_t)B * T);
    V100_GGL(ctc_lse_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, logits, lse, rows, V, stall, B);
    const size_t shmem = (size_t)(CTC_CHUNK * V + 2 * (Smax + 4)) * sizeof(float);
#define CTC_LATTICE(NS_)                                                                                                          \
    if (shmem > 65536)                                                                                                            \
        (void)hipFuncSetAttribute((const void*)ctc_lattice_kernel<NS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);       \
    V100_GGL(ctc_lattice_kernel<NS_>, dim3(B, 2), dim3(256), shmem, st, logits, lse, targets, in_len, tgt_len, alpha, beta, \
                       nll, T, V, Lmax, Smax, blank)
    // Round 6: transcripts of up to 127 tokens take the linear-domain fp64 pair first; the log-domain pair behind it then runs only for
    // the utterances whose per-frame mass check failed (`only`), and forms the 'mean'
    const int* only = nullptr;
    if (ctc_lin_enabled() && Smax <= CTCL_MAXS) {
        float* tail = workspace + ctc_ws_log(B, T, Lmax);
        double* ll2d = (double*)(((size_t)tail + 7) & ~(size_t)7);
        float* lat = (float*)(ll2d + B);                              // (16-byte aligned rows: CTCL_ROW * 4 bytes is a multiple of 16)
        lat = (float*)(((size_t)lat + 15) & ~(size_t)15);
        int* bad = (int*)(lat + 2 * (size_t)B * T * CTCL_ROW);
        int chunk = (int)(49152 / ((size_t)(V + 1) * sizeof(double))) - 1;        // two buffers of <= 48 KB (+ one spare row each)
        if (chunk > 128) chunk = 128;                                             // (short first chunk: its staging is not hidden)
        if (chunk > T) chunk = T;
        if (chunk < 1) return V100_ERR_SHAPE;
        const size_t lds = 2 * (size_t)(chunk + 1) * (V + 1) * sizeof(double);
        if (lds > 65536) {
            static bool raised = false;
            if (!raised) { (void)hipFuncSetAttribute((const void*)ctc_lattice_lin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 49152 + 4096); raised = true; }
        }
        V100_GGL(ctc_lattice_lin_kernel, dim3(B, 2), dim3(256), lds, st, logits, lse, targets, in_len, tgt_len, lat, nll, ll2d, bad,
                 B, T, V, Lmax, blank, chunk);
        V100_GGL(ctc_grad_lin_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, logits, lse, targets, in_len, tgt_len, lat,
                 nll, ll2d, grad, bad, B, T, V, Lmax, blank, loss ? 1 : 0);
        only = bad;
    }
    if (CTC_SKEW && Smax <= 1024) {
        const int nw = (Smax + 63) / 64;
        V100_GGL(ctc_lattice_skew_kernel, dim3(B, 2), dim3(64 * nw), 0, st, logits, lse, targets, in_len, tgt_len, alpha, beta,
                           nll, T, V, Lmax, Smax, blank, stall, only);
    } else if (Smax <= 512) { CTC_LATTICE(2); }
    else if (Smax <= 1024) { CTC_LATTICE(4); }
    else if (Smax <= 2048) { CTC_LATTICE(8); }
    else { CTC_LATTICE(16); }
#undef CTC_LATTICE
    V100_GGL(ctc_grad_kernel, dim3((unsigned)(((only ? (long)T : rows) + 3) / 4)), dim3(256), 0, st, logits, lse, targets, in_len, tgt_len,
                       alpha, beta, nll, grad, B, T, V, Lmax, Smax, blank, loss ? 1 : 0, stall, loss, only);     // (the 'mean' reduction rides in block 0)
    return v100_launch_status();
}
