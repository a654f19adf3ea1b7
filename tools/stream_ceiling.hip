// What a plain streaming kernel with kf_xp_Ax's access mix reaches on this GPU: the yardstick for the 4K CG kernel's HBM fraction (Poisson
// section).  Per float4 element: NR coalesced dwordx4 loads from NR arrays, NW dwordx4 stores (plain or non-temporal) to NW other arrays, no reuse,
// arrays of the 4K solver's size (3840 x 2160 x 3 floats = 99.5 MB each).  kf_xp_Ax reads x, r, p and writes x, p_new, Ap: NR = 3, NW = 3 (72 B/px).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/stream_ceiling tools/stream_ceiling.hip ;  gpurun -- 'tools/bin/stream_ceiling'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NR, int NW, bool NT>
__global__ __launch_bounds__(256) void k_stream(const float4 *const *in, float4 *const *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 s = {0, 0, 0, 0};
#pragma unroll
        for (int a = 0; a < NR; a++) { const float4 v = in[a][i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
#pragma unroll
        for (int a = 0; a < NW; a++) {
            float4 v = s; v.x += (float)a;
            if (NT) { __builtin_nontemporal_store(v.x, &out[a][i].x); __builtin_nontemporal_store(v.y, &out[a][i].y); __builtin_nontemporal_store(v.z, &out[a][i].z); __builtin_nontemporal_store(v.w, &out[a][i].w); }
            else out[a][i] = v;
        }
    }
}
template <int NR, int NW, bool NT>
void run(const float4 *const *din, float4 *const *dout, size_t n, int grid)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 12; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_stream<NR, NW, NT>), dim3(grid), dim3(256), 0, 0, din, dout, n);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    const double bytes = (double)(NR + NW) * n * 16.0;
    printf("reads %d writes %d %s grid %6d: %7.1f us  %5.2f TB/s\n", NR, NW, NT ? "nt-stores" : "stores   ", grid, best * 1e3, bytes / (best * 1e-3) / 1e12);
}
int main()
{
    const size_t n = (size_t)3840 * 2160 * 3 / 4;
    std::vector<float4 *> bufs(8);
    for (auto &b : bufs) { hipMalloc(&b, n * 16); hipMemset(b, 0, n * 16); }
    const float4 **din; float4 **dout;
    hipMalloc(&din, 4 * sizeof(void *)); hipMalloc(&dout, 4 * sizeof(void *));
    hipMemcpy(din, bufs.data(), 4 * sizeof(void *), hipMemcpyHostToDevice);
    hipMemcpy(dout, bufs.data() + 4, 4 * sizeof(void *), hipMemcpyHostToDevice);
    for (int grid : {1024, 2048, 4096, 8192, 24300}) {
        run<1, 1, false>(din, dout, n, grid); run<1, 1, true>(din, dout, n, grid);
        run<3, 3, false>(din, dout, n, grid); run<3, 3, true>(din, dout, n, grid);
        run<2, 1, true>(din, dout, n, grid); run<4, 0 + 1, true>(din, dout, n, grid);
    }
    for (int grid : {4096}) { run<1, 0 + 1, true>(din, dout, n, grid); }
    return 0;
}
