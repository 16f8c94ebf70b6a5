/* Dense attention forward/backward in double precision, plain C: an independent CPU restatement of
 * /root/reference/src/common/correctness.py:5-34 (reference_attention / reference_backward:
 * scores = QK^T*scale, causal mask col > row -> -inf per src/common/mask.py:6-12, softmax, PV,
 * lse = logsumexp; backward = analytic gradient of that function).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/attention_oracle.py header): loaded by tests/ through ctypes to
 * cross-check the torch-based oracle with code that shares nothing with torch. Parity status: pinned
 * (tests/test_oracle_golden.py::test_c_oracle_matches_reference_vectors).
 *
 * q,k,v,do: (bh, n, d) float; out: o (bh,n,d) float, lse (bh,n) float, dq,dk,dv (bh,n,d) float (may be NULL
 * together with do to skip the backward). Returns 0, or -1 on allocation failure. */
#include <math.h>
#include <stdlib.h>

int attn_ref_f64(const float* q, const float* k, const float* v, const float* dout, long bh, long n, long d,
                 int causal, double scale, float* o, float* lse, float* dq, float* dk, float* dv) {
    double* p = (double*)malloc(sizeof(double) * (size_t)n);
    double* orow = (double*)malloc(sizeof(double) * (size_t)d);
    double* acc_dk = dout ? (double*)calloc((size_t)n * (size_t)d, sizeof(double)) : NULL;
    double* acc_dv = dout ? (double*)calloc((size_t)n * (size_t)d, sizeof(double)) : NULL;
    if (!p || !orow || (dout && (!acc_dk || !acc_dv))) return -1;
    for (long b = 0; b < bh; ++b) {
        const float *qb = q + b * n * d, *kb = k + b * n * d, *vb = v + b * n * d;
        if (dout) for (long i = 0; i < n * d; ++i) { acc_dk[i] = 0.0; acc_dv[i] = 0.0; }
        for (long i = 0; i < n; ++i) {
            const long lim = causal ? i + 1 : n; /* keys j > i are masked */
            double mx = -INFINITY;
            for (long j = 0; j < lim; ++j) {
                double s = 0.0;
                for (long c = 0; c < d; ++c) s += (double)qb[i * d + c] * (double)kb[j * d + c];
                p[j] = s * scale;
                if (p[j] > mx) mx = p[j];
            }
            double sum = 0.0;
            for (long j = 0; j < lim; ++j) { p[j] = exp(p[j] - mx); sum += p[j]; }
            for (long j = 0; j < lim; ++j) p[j] /= sum;
            for (long c = 0; c < d; ++c) orow[c] = 0.0;
            for (long j = 0; j < lim; ++j)
                for (long c = 0; c < d; ++c) orow[c] += p[j] * (double)vb[j * d + c];
            for (long c = 0; c < d; ++c) o[(b * n + i) * d + c] = (float)orow[c];
            lse[b * n + i] = (float)(mx + log(sum));
            if (dout) {
                const float* dor = dout + (b * n + i) * d;
                double delta = 0.0;
                for (long c = 0; c < d; ++c) delta += (double)dor[c] * orow[c];
                for (long c = 0; c < d; ++c) orow[c] = 0.0; /* reuse as dq row */
                for (long j = 0; j < lim; ++j) {
                    double dp = 0.0;
                    for (long c = 0; c < d; ++c) dp += (double)dor[c] * (double)vb[j * d + c];
                    const double ds = p[j] * (dp - delta) * scale;
                    for (long c = 0; c < d; ++c) {
                        orow[c] += ds * (double)kb[j * d + c];
                        acc_dk[j * d + c] += ds * (double)qb[i * d + c];
                        acc_dv[j * d + c] += p[j] * (double)dor[c];
                    }
                }
                for (long c = 0; c < d; ++c) dq[(b * n + i) * d + c] = (float)orow[c];
            }
        }
        if (dout) for (long i = 0; i < n * d; ++i) { dk[b * n * d + i] = (float)acc_dk[i]; dv[b * n * d + i] = (float)acc_dv[i]; }
    }
    free(p); free(orow); free(acc_dk); free(acc_dv);
    return 0;
}
