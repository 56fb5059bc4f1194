// Probe: what does `buffer_load_dwordx4 ... lds` leave in LDS for lanes whose offset is outside the buffer's range?
// (K10's halo pixels outside the image want zeros there.)   hipcc --offload-arch=gfx950 -O2 lds_dma_oob.hip -o lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned* src, unsigned nbytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xFFFFFFFFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, nbytes, 0x00020000);
    const int lane = threadIdx.x;
    const int voff = (lane & 1) ? 0x7FFFFFF0 : lane * 16;      // odd lanes: far outside the range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
    unsigned *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, 1024, o);
    std::vector<unsigned> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 8; ++l) printf("lane %d: %08x %08x %08x %08x\n", l, r[4 * l], r[4 * l + 1], r[4 * l + 2], r[4 * l + 3]);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { unsigned want = (l & 1) ? 0u : 0x1000u + 4 * l + j; if (r[4 * l + j] != want) ++bad; }
    printf("mismatches against 'in range: data, out of range: zeros' = %d\n", bad);
    return 0;
}
