// Prints the lane mapping of the DPP / permlane controls used by gf_reduce_scatter32 (run on gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned* out) {
    const unsigned l = threadIdx.x;
    out[0 * 64 + l] = __builtin_amdgcn_update_dpp(999u, l, 0x128, 0xF, 0xF, true);    // row_ror:8
    out[1 * 64 + l] = __builtin_amdgcn_update_dpp(999u, l, 0x124, 0xF, 0xF, true);    // row_ror:4
    out[2 * 64 + l] = __builtin_amdgcn_update_dpp(999u, l, 0x12C, 0xF, 0xF, true);    // row_ror:12
    out[3 * 64 + l] = __builtin_amdgcn_update_dpp(999u, l, 0x12C, 0xF, 0x5, false);   // row_ror:12 banks 0,2 (old=999)
    out[4 * 64 + l] = __builtin_amdgcn_update_dpp(999u, l, 0xB1, 0xF, 0xF, true);     // quad_perm 1,0,3,2
    out[5 * 64 + l] = __builtin_amdgcn_update_dpp(999u, l, 0x4E, 0xF, 0xF, true);     // quad_perm 2,3,0,1
    v2u r = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
    out[6 * 64 + l] = r.x;
    out[7 * 64 + l] = r.y;
}
int main() {
    unsigned* d; unsigned h[8 * 64];
    hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"row_ror:8", "row_ror:4", "row_ror:12", "row_ror:12 bank0x5", "quad_perm 1032", "quad_perm 2301", "swap16 vdst", "swap16 src"};
    for (int k = 0; k < 8; ++k) { printf("%-20s", names[k]); for (int l = 0; l < 36; ++l) printf(" %3u", h[k * 64 + l]); printf("\n"); }
    return 0;
}
