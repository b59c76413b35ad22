/* WORLD's randn() sequence (TEST INFRASTRUCTURE; parity unpinned, see oracle/world_synth.py): the value after randn_reseed() is the
 * sum of twelve draws of a 32-bit xorshift128 generator (seeds 123456789, 362436069, 521288629, 88675123), each shifted right by 4,
 * scaled by 2^-28, minus 6 -- an approximately normal variate.  Fills out[0..n). */
#include <stdint.h>
void world_randn_fill(double* out, long n) {
    uint32_t x = 123456789u, y = 362436069u, z = 521288629u, w = 88675123u;
    for (long i = 0; i < n; ++i) {
        uint32_t tmp = 0;
        for (int j = 0; j < 12; ++j) {
            uint32_t t = x ^ (x << 11);
            x = y; y = z; z = w;
            w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
            tmp += w >> 4;
        }
        out[i] = tmp / 268435456.0 - 6.0;
    }
}
