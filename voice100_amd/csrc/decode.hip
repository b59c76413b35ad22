// the integer steps right after / between the CNNs, on the device, bit-exact.
//
//   ctc_greedy_decode   argmax per frame, collapse repeats, drop blanks  (CharTokenizer.merge_repeated on the
//                       decoded string, voice100/text.py:99-104; argmax as in export_onnx / inference glue)
//   ctc_best_path       forced alignment of a label sequence to frame log-probabilities: banded Viterbi over the
//                       blank-expanded labels, at most max_move-1 positions skipped per frame, a 2-step move may not
//                       land on a blank, ties resolve to the smallest move (voice100/models/align.py:18-66)
//   align_expand        TextToAlignTextModel.align: lay token i over [round(t+gap_i), round(t+gap_i+len_i)) with
//                       Python round() semantics (banker's rounding of the running double), >= 1 frame per token,
//                       head/tail blank frames (voice100/models/tts.py:89-110)
#include "common.h"
#include <math.h>

// one workgroup per utterance; T <= 65536
__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float* __restrict__ logits, const int* __restrict__ lens,
                                                         long long* __restrict__ out, int* __restrict__ out_len, int T, int V, int blank) {
    extern __shared__ int sm[];                 // [T] argmax ids, then reused as keep flags / positions
    __shared__ int wsum[4];
    __shared__ int carry;
    const int b = blockIdx.x, tid = threadIdx.x;
    int Tb = lens ? lens[b] : T;
    if (Tb > T) Tb = T;
    for (int t = tid; t < Tb; t += 256) {
        const float* p = logits + ((size_t)b * T + t) * V;
        float best = p[0];
        int arg = 0;
        for (int c = 1; c < V; ++c) { const float v = p[c]; if (v > best) { best = v; arg = c; } }   // first maximum wins (torch.argmax)
        sm[t] = arg;
    }
    if (tid == 0) carry = 0;
    __syncthreads();
    // ordered compaction, 256 frames per round
    for (int base = 0; base < Tb; base += 256) {
        const int t = base + tid;
        int keep = 0, id = 0;
        if (t < Tb) {
            id = sm[t];
            keep = (id != blank) && (t == 0 || sm[t - 1] != id);
        }
        // exclusive prefix sum of keep over the block
        const int lane = tid & 63, wave = tid >> 6;
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (keep) out[(size_t)b * T + off + before] = id;
        __syncthreads();
        if (tid == 0) carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) out_len[b] = carry;
    for (int t = carry + tid; t < T; t += 256) out[(size_t)b * T + t] = 0;
}

// one workgroup per utterance. logp [B][T][V] fp32 log-probabilities, labels [B][Lmax] int64, lens.
// back [B][T][S] int16 workspace, path [B][T] int32 out, score [B] out.  S = 2*Lmax+1 <= 1024.
__global__ __launch_bounds__(256) void ctc_best_path_kernel(const float* __restrict__ logp, const long long* __restrict__ labels,
                                                            const int* __restrict__ in_len, const int* __restrict__ lab_len,
                                                            short* __restrict__ back, int* __restrict__ path, float* __restrict__ score,
                                                            int T, int V, int Lmax, int Smax, int max_move) {
    extern __shared__ float sc[];               // [2][Smax] scores
    const int b = blockIdx.x, tid = threadIdx.x;
    int Tb = in_len ? in_len[b] : T;
    if (Tb > T) Tb = T;
    int L = lab_len ? lab_len[b] : Lmax;
    if (L > Lmax) L = Lmax;
    const int n = 2 * L + 1;
    const long long* lab = labels + (size_t)b * Lmax;
    const float* lp = logp + (size_t)b * T * V;
    short* bk = back + (size_t)b * T * Smax;
    auto ext = [&](int v) -> int { return (v & 1) ? (int)lab[v >> 1] : 0; };
    // frame 0: positions 0 and 1 are live (align.py:30)
    int width = n < 2 ? n : 2;
    for (int v = tid; v < Smax; v += 256) { sc[v] = (v < width) ? lp[ext(v)] : -INFINITY; }
    __syncthreads();
    for (int i = 1; i < Tb; ++i) {
        const float* cur = sc + ((i - 1) & 1) * Smax;
        float* nxt = sc + (i & 1) * Smax;
        const int nwidth = min(width + max_move - 1, n);
        for (int v = tid; v < nwidth; v += 256) {
            const int e = ext(v);
            const float emit = lp[(size_t)i * V + e];
            float best = -INFINITY;
            int arg = 0;
            bool first = true;
            for (int j = 0; j < max_move; ++j) {
                const int k = v - j;
                float cand = -INFINITY;
                int src = 0;
                if (k >= 0 && k < width) {
                    cand = cur[k] + emit;
                    if (j > 0 && (j & 1) == 0 && e == 0) cand = -INFINITY;      // no blank -> blank jump
                    src = k;
                }
                if (first || cand > best) { best = cand; arg = src; first = false; }   // np.argmax: first maximum
            }
            nxt[v] = best;
            bk[(size_t)i * Smax + v] = (short)arg;
        }
        __syncthreads();
        width = nwidth;
    }
    if (tid == 0) {
        const float* fin = sc + ((Tb - 1) & 1) * Smax;
        int j = (width >= 2 && fin[width - 1] > fin[width - 2]) ? n - 1 : n - 2;     // align.py:58 (uses the last two live scores)
        if (j < 0) j = 0;
        if (j >= width) j = width - 1;
        score[b] = fin[j];
        for (int i = Tb - 1; i >= 0; --i) {
            path[(size_t)b * T + i] = j;
            if (i > 0) j = bk[(size_t)i * Smax + j];
        }
        for (int i = Tb; i < T; ++i) path[(size_t)b * T + i] = 0;
    }
}

// one workgroup per utterance: thread 0 walks the serial double-precision prefix sum (L is a few hundred at most) and leaves each
// token's [s, e) span in LDS / global scratch-free form: spans are written to the output row by ALL threads afterwards.  Exactly the
// reference's loop: later tokens overwrite earlier ones where spans overlap (they can, through rounding), so the fill walks tokens in
// order per output position: position j takes the LAST token i whose span covers it.
__global__ __launch_bounds__(256) void align_expand_kernel(const long long* __restrict__ text, const double* __restrict__ align,
                                                           const int* __restrict__ text_len, long long* __restrict__ out,
                                                           int* __restrict__ out_len, int Lmax, int Tmax, int head, int tail) {
    extern __shared__ int spans[];                 // [2 * Lmax]: s_i, e_i
    __shared__ int n_sh;
    const int b = blockIdx.x;
    const int L = text_len ? min(text_len[b], Lmax) : Lmax;
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int i = 0; i < L; ++i) total += align[((size_t)b * Lmax + i) * 2] + align[((size_t)b * Lmax + i) * 2 + 1];
        int n = head + (int)total + tail;            // int(torch.sum(align)) truncates toward zero
        if (out && n > Tmax) n = Tmax;
        double t = (double)head;
        for (int i = 0; i < L; ++i) {
            t += align[((size_t)b * Lmax + i) * 2];
            int s = (int)rint(t);                    // Python round(): half to even on the double
            t += align[((size_t)b * Lmax + i) * 2 + 1];
            int e = (int)rint(t);
            if (s == e) e = e + 1 > 0 ? e + 1 : 0;
            if (s < 0) s = 0;                        // python range(s, e) with negative s would index from the end; never happens with head >= 0 and gaps >= 0
            spans[2 * i] = s;
            spans[2 * i + 1] = e < n ? e : n;
        }
        n_sh = n;
        out_len[b] = n;
    }
    __syncthreads();
    if (!out) return;                                // length query only
    const int n = n_sh;
    long long* o = out + (size_t)b * Tmax;
    for (int j = threadIdx.x; j < Tmax; j += blockDim.x) {
        long long v = 0;
        if (j < n) {
            // spans are (nearly) sorted: scan backwards from the end for the last covering token -- L is small, and the loop is
            // uniform enough (every thread scans the same list)
            for (int i = L - 1; i >= 0; --i)
                if (j >= spans[2 * i] && j < spans[2 * i + 1]) { v = text[(size_t)b * Lmax + i]; break; }
        }
        o[j] = v;
    }
}

extern "C" int v100_ctc_greedy_decode(const float* logits, const int* lens, long long* out, int* out_len, int B, int T, int V, int blank, void* stream) {
    if (!logits || !out || !out_len) return V100_ERR_NULL;
    if (B <= 0 || T <= 0 || T > 12000 || V <= 0 || blank < 0 || blank >= V) return V100_ERR_SHAPE;
    V100_GGL(ctc_greedy_kernel, dim3(B), dim3(256), T * sizeof(int), (hipStream_t)stream, logits, lens, out, out_len, T, V, blank);
    return v100_launch_status();
}

extern "C" int v100_ctc_best_path(const float* logp, const long long* labels, const int* in_len, const int* lab_len, void* back_ws, int* path,
                                  float* score, int B, int T, int V, int Lmax, int max_move, void* stream) {
    if (!logp || !labels || !back_ws || !path || !score) return V100_ERR_NULL;
    const int Smax = 2 * Lmax + 1;
    if (B <= 0 || T <= 0 || V <= 0 || Lmax < 0 || Smax > 4096 || max_move < 1 || max_move > 8) return V100_ERR_SHAPE;
    V100_GGL(ctc_best_path_kernel, dim3(B), dim3(256), 2 * Smax * sizeof(float), (hipStream_t)stream, logp, labels, in_len, lab_len,
                       (short*)back_ws, path, score, T, V, Lmax, Smax, max_move);
    return v100_launch_status();
}

extern "C" int v100_align_expand(const long long* text, const double* align, const int* text_len, long long* out, int* out_len, int B, int Lmax,
                                 int Tmax, int head, int tail, void* stream) {
    if (!text || !align || !out_len) return V100_ERR_NULL;            // out == NULL: only the lengths are written (the caller sizes `out` from them)
    if (B <= 0 || Lmax <= 0 || Lmax > 8192 || (out && Tmax <= 0) || head < 0 || tail < 0) return V100_ERR_SHAPE;
    V100_GGL(align_expand_kernel, dim3(B), dim3(256), 2 * Lmax * sizeof(int), (hipStream_t)stream, text, align, text_len, out, out_len, Lmax,
                       Tmax, head, tail);
    return v100_launch_status();
}
