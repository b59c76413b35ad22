/* Plain-C restatement of the layer arithmetic on the Voice100 hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): an independent, loop-level definition of
 * Conv1d / ConvTranspose1d / BatchNorm1d / ReLU6 used to cross-check both the torch-based oracle
 * (oracle/cnn.py) and the HIP kernels on small shapes.  Accumulates in double.
 *
 * Reference semantics restated:
 *   nn.Conv1d(Cin, Cout, k, stride, padding, groups, bias)   voice100/models/asr.py:31-35, 51, 91
 *   nn.BatchNorm1d (train: biased batch variance; eval: running stats), eps 1e-5   asr.py:36, 52
 *   nn.ReLU6                                                  asr.py:37
 *   nn.ConvTranspose1d(Cin, Cout, k=5, padding=2, stride=2)   voice100/models/tts.py:22
 */
#include <math.h>
#include <stddef.h>

/* y[b][co][t] = bias[co] + sum_{ci in group, j} w[co][ci_local][j] * x[b][ci][t*stride - pad + j] */
void ref_conv1d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout,
                int T, int K, int stride, int pad, int groups) {
    const int Tout = (T + 2 * pad - K) / stride + 1;
    const int cin_g = Cin / groups, cout_g = Cout / groups;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co) {
            const int g = co / cout_g;
            for (int t = 0; t < Tout; ++t) {
                double acc = bias ? bias[co] : 0.0;
                for (int cl = 0; cl < cin_g; ++cl) {
                    const int ci = g * cin_g + cl;
                    for (int j = 0; j < K; ++j) {
                        const int ti = t * stride - pad + j;
                        if (ti < 0 || ti >= T) continue;
                        acc += (double)w[((size_t)co * cin_g + cl) * K + j] * (double)x[((size_t)b * Cin + ci) * T + ti];
                    }
                }
                y[((size_t)b * Cout + co) * Tout + t] = (float)acc;
            }
        }
}

/* y[b][co][to] = bias[co] + sum_{ci, j : to = ti*stride - pad + j} w[ci][co][j] * x[b][ci][ti];  Tout = (T-1)*stride - 2*pad + K */
void ref_conv_transpose1d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout,
                          int T, int K, int stride, int pad) {
    const int Tout = (T - 1) * stride - 2 * pad + K;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int to = 0; to < Tout; ++to) {
                double acc = bias ? bias[co] : 0.0;
                for (int ci = 0; ci < Cin; ++ci)
                    for (int j = 0; j < K; ++j) {
                        const int num = to + pad - j;
                        if (num < 0 || num % stride) continue;
                        const int ti = num / stride;
                        if (ti >= T) continue;
                        acc += (double)w[((size_t)ci * Cout + co) * K + j] * (double)x[((size_t)b * Cin + ci) * T + ti];
                    }
                y[((size_t)b * Cout + co) * Tout + to] = (float)acc;
            }
}

/* training=1: batch mean / biased variance over (B,T), also returned; training=0: uses mean[]/var[] as given */
void ref_batchnorm1d(const float* x, const float* gamma, const float* beta, float* mean, float* var, float* y,
                     int B, int C, int T, float eps, int training) {
    for (int c = 0; c < C; ++c) {
        double mu, v;
        if (training) {
            double s = 0.0, s2 = 0.0;
            for (int b = 0; b < B; ++b)
                for (int t = 0; t < T; ++t) s += x[((size_t)b * C + c) * T + t];
            mu = s / ((double)B * T);
            for (int b = 0; b < B; ++b)
                for (int t = 0; t < T; ++t) {
                    const double d = x[((size_t)b * C + c) * T + t] - mu;
                    s2 += d * d;
                }
            v = s2 / ((double)B * T);
            mean[c] = (float)mu;
            var[c] = (float)v;
        } else {
            mu = mean[c];
            v = var[c];
        }
        const double rs = 1.0 / sqrt(v + (double)eps);
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < T; ++t) {
                const size_t i = ((size_t)b * C + c) * T + t;
                y[i] = (float)((x[i] - mu) * rs * gamma[c] + beta[c]);
            }
    }
}

void ref_relu6(float* x, size_t n) {
    for (size_t i = 0; i < n; ++i) x[i] = x[i] < 0.f ? 0.f : (x[i] > 6.f ? 6.f : x[i]);
}
