// Development probe: what ds_read_b64_tr_b16 returns.  lds[i] = i (16-bit); every lane reads "its" 8 bytes at lane * 8
// (natural addressing) -> prints element j of lane l.  Expectation (cdna_hip_programming.md section 2):
//   lane l, element j  ==  lds[(l & 15) + 16 j + 64 (l >> 4)]
// Second pass: row stride 32 B with the rows of a 16-lane group at arbitrary positions (the layout edge_agg.hip uses).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int mode) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, j = (l & 15) >> 2, c = l & 3;
    int off;                       // in 16-bit elements
    if (mode == 0) off = l * 4;
    else {                         // row r = 8 g + j of a [32 rows][16 ch] tile, row r stored at position pos(r) (bits 2/3 swapped)
        const int r = 8 * g + j + (mode == 2 ? 4 : 0);
        const int pos = (r & 3) | (((r >> 3) & 1) << 2) | (((r >> 2) & 1) << 3) | (r & 16);
        off = pos * 16 + c * 4;
    }
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + off));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
    short* d;
    hipMalloc(&d, 64 * 4 * 2);
    short h[256];
    int bad = 0;
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                int want;
                if (mode == 0) want = (l & 15) + 16 * e + 64 * (l >> 4);
                else {             // expected: row 8 g + e (+4) at channel l & 15 -> stored at pos(row) * 16 + channel
                    const int r = 8 * (l >> 4) + e + (mode == 2 ? 4 : 0);
                    const int pos = (r & 3) | (((r >> 3) & 1) << 2) | (((r >> 2) & 1) << 3) | (r & 16);
                    want = pos * 16 + (l & 15);
                }
                if (h[l * 4 + e] != want) {
                    if (bad < 16) printf("mode %d lane %d elem %d: got %d want %d\n", mode, l, e, h[l * 4 + e], want);
                    ++bad;
                }
            }
    }
    printf("tr16 probe: %d mismatches\n", bad);
    return bad != 0;
}
