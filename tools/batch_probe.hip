// scratch probe: ablation of vg_batch_kernel<48> (what costs the non-MFMA cycles?).  Build per variant:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVGB_ABLATE=<bits> -I../sqlite-vector_amd/csrc batch_probe.hip -o batch_probe_<bits>
#include "../sqlite-vector_amd/csrc/vg_batch.hip"
#include <cstdio>
#include <vector>
int main() {
    const long long n = 2000000; const int dim = 384, nq = 1024, k = 20;
    float *rows, *q; uint64_t *cand, *out;
    hipMalloc(&rows, n * dim * 4); hipMalloc(&q, (size_t)nq * dim * 4);
    std::vector<float> h((size_t)1 << 20);
    for (auto &x : h) x = (float)rand() / RAND_MAX - 0.5f;
    for (long long off = 0; off < n * dim; off += (1 << 20)) hipMemcpy(rows + off, h.data(), std::min<long long>(1 << 20, n * dim - off) * 4, hipMemcpyHostToDevice);
    hipMemcpy(q, h.data(), (size_t)nq * dim * 4, hipMemcpyHostToDevice);
    const int npart = 32; const long long ntiles = (n + 31) / 32; const int tpp = (int)((ntiles + npart - 1) / npart);
    hipMalloc(&cand, (size_t)nq * npart * 64 * 8); hipMalloc(&out, (size_t)nq * 64 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        int rc = vg_batch_launch(rows, n, dim * 4, q, nq, k, 0, cand, npart, tpp, out, 0);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("ablate=%d rc=%d  %.3f ms  %.1f TFLOP/s\n", VGB_ABLATE, rc, ms, 2.0 * nq * n * dim / (ms * 1e-3) / 1e12);
    }
    return 0;
}
