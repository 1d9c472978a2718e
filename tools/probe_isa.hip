// Developer probe: dumps the lane<->element maps of ds_read_b64_tr_b16 and
// v_mfma_f32_16x16x32_bf16 on the GPU it runs on (used once to confirm the
// fragment layouts attn.hip relies on).  hipcc --offload-arch=gfx950 probe_isa.hip -o probe_isa
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe_tr(float* out) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[1024];
    int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = (__bf16)(float)i;  // exact up to 256; use i%256 + block tag
    __syncthreads();
    // lane l points at elements [4l, 4l+4)
    bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds + 4 * l));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)t[j];
}
// A[i][k] = i*32+k (as small ints via two passes), B = one-hot to read layouts
__global__ void probe_mfma(float* out) {
    int l = threadIdx.x;
    // pass 1: A = indicator(row == r0, k == k0) for lane's own slots -> find which (i,k) each lane slot is:
    // use B = all ones: D[i][j] = sum_k A[i][k]. Set A slot t of lane l = 2^-(t) * (l+1)?  Simpler: numeric encode.
    // D[i][j] with A[i][k] = (i==I)&&(k==K) ? 1 : 0 would need 512 launches; instead encode:
    //   A value held by lane l slot t := l*8+t (exact in bf16 only < 256) -> use two MFMAs with hi/lo parts.
    bf16x8 a_lo, a_hi, b;
    for (int t = 0; t < 8; ++t) {
        int v = l * 8 + t;           // 0..511
        a_lo[t] = (__bf16)(float)(v & 15);
        a_hi[t] = (__bf16)(float)(v >> 4);
        b[t] = (__bf16)0.0f;
    }
    // B one-hot at k = kk (runtime loop): B[k][j] = (k==kk). lane l slot t holds B[k=(l>>4)*8+t][j=l&15] (assumed);
    // we do not assume: set B slot (l,t) = 1 iff (l>>4)*8+t == kk under the ASSUMED layout and report what comes out;
    // consistency of the result with the assumed A layout verifies both.
    for (int kk = 0; kk < 32; ++kk) {
        for (int t = 0; t < 8; ++t) b[t] = (__bf16)((((l >> 4) * 8 + t) == kk) ? 1.0f : 0.0f);
        f32x4 z = {0, 0, 0, 0};
        f32x4 dlo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b, z, 0, 0, 0);
        f32x4 dhi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b, z, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[(kk * 64 + l) * 4 + r] = dhi[r] * 16.0f + dlo[r];
    }
}
int main() {
    float *d, h[32 * 64 * 4];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 64 * 4 * 4, hipMemcpyDeviceToHost);
    printf("ds_read_b64_tr_b16: lane l addr=elements[4l..4l+3]; result elems per lane:\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4.0f %4.0f %4.0f %4.0f   expect %d %d %d %d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3],
        (l & 15) + 0 * 16 + (l >> 4) * 64, (l & 15) + 16 + (l >> 4) * 64, (l & 15) + 32 + (l >> 4) * 64, (l & 15) + 48 + (l >> 4) * 64);
    hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // D[i][j] for one-hot B at k=kk equals A[i][kk] (for all j). Under assumed layouts A[i][k] is held by lane (i + 16*(k/8)) slot k%8
    // -> value (i+16*(k/8))*8 + k%8; D lane l reg r holds D[i=(l>>4)*4+r][j=l&15].
    int bad = 0;
    for (int kk = 0; kk < 32; ++kk)
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                int i = (l >> 4) * 4 + r;
                float expect = (float)((i + 16 * (kk / 8)) * 8 + kk % 8);
                if (h[(kk * 64 + l) * 4 + r] != expect) {
                    if (bad < 20) printf("mfma mismatch kk=%d lane=%d r=%d got %.0f expect %.0f\n", kk, l, r, h[(kk * 64 + l) * 4 + r], expect);
                    ++bad;
                }
            }
    printf("mfma layout check: %s (%d mismatches)\n", bad ? "MISMATCH" : "as assumed", bad);
    return 0;
}
