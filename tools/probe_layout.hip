// tools/probe_layout.hip — prints the gfx950 lane layouts this kernel relies on, from the
// hardware itself: (1) v_mfma_f32_32x32x16_bf16 C/D layout and A/B row/col ownership,
// (2) ds_read_b64_tr_b16 transpose semantics.  Build: hipcc --offload-arch=gfx950 -O2 probe_layout.hip -o probe_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe_mfma(float* out) {
  // A[i][k] = (i == I0 && k-slot of this lane) ... use A = one-hot rows, B = one-hot cols.
  // Lane l supplies A row (l&31), 8 k-values; B col (l&31), 8 k-values.  Set A[i][k] = i+1 for
  // k position 0 of half 0 only, B[k][j] = 100*(j+1) for the same k: then C[i][j] = (i+1)*100*(j+1).
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)0.f; b[e] = (__bf16)0.f; }
  if (l < 32) { a[0] = (__bf16)(float)(l + 1); b[0] = (__bf16)(float)(l + 1); }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}

__global__ void probe_kperm(float* out) {
  // Does the k index pair up element-for-element between A and B for BOTH half-waves?
  // A[i][*] = 1 in element e of half h only; B[*][j] = 1 in element e2 of half h2 only.
  // C should be 32*... nonzero only when (e,h)==(e2,h2).  out[(h*8+e)*16 + (h2*8+e2)] = C[0][0].
  int l = threadIdx.x;
  for (int ah = 0; ah < 2; ++ah) for (int ae = 0; ae < 8; ++ae)
    for (int bh = 0; bh < 2; ++bh) for (int be = 0; be < 8; ++be) {
      bf16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (__bf16)0.f; b[e] = (__bf16)0.f; }
      if ((l >> 5) == ah) a[ae] = (__bf16)1.f;
      if ((l >> 5) == bh) b[be] = (__bf16)1.f;
      f32x16 c = {};
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
      if (l == 0) out[(ah * 8 + ae) * 16 + (bh * 8 + be)] = c[0];
    }
}

__global__ void probe_tr(short* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[2048];
  int l = threadIdx.x;
  for (int i = l; i < 2048; i += 64) lds[i] = (short)i;
  __syncthreads();
  // mode 0: lane-linear addresses (lane l -> halfwords 4l..4l+3)
  // mode 1: the V-tile addressing of the kernel: group g=(l>>4)&1, i=l&15, hi=l>>5:
  //         byte = hi*512 + (i>>2)*64 + g*32 + (i&3)*8   -> expect lane gets column (l&31) of rows 0..3 of a [8][32] tile
  int byte = (mode == 0) ? l * 8 : ((l >> 5) * 512 + ((l & 15) >> 2) * 64 + ((l >> 4) & 1) * 32 + (l & 3) * 8);
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + byte));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = t[e];
}

int main() {
  float* d; hipMalloc(&d, 64 * 16 * 4 + 256 * 4);
  float h[64 * 16];
  probe_mfma<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("== mfma_f32_32x32x16_bf16 C layout: value = (row+1)*(col+1); print lane,reg -> row,col\n");
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float want = (float)((row + 1) * (col + 1));
    if (h[l * 16 + r] != want) { if (bad < 8) printf("  MISMATCH lane %d reg %d got %g want %g\n", l, r, h[l * 16 + r], want); bad++; }
  }
  printf("  C layout [col=lane&31,row=(r&3)+8*(r>>2)+4*(lane>>5)] with A row=lane&31, B col=lane&31: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  float hk[256];
  probe_kperm<<<1, 64>>>(d);
  hipMemcpy(hk, d, sizeof(hk), hipMemcpyDeviceToHost);
  bad = 0;
  for (int a = 0; a < 16; ++a) for (int b = 0; b < 16; ++b) { float want = (a == b) ? 1.f : 0.f; if (hk[a * 16 + b] != want) bad++; }
  printf("== k pairing: A(half,elem) contracts only with B(same half, same elem): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  short* ds; hipMalloc(&ds, 64 * 4 * 2);
  short hs[256];
  for (int mode = 0; mode < 2; ++mode) {
    probe_tr<<<1, 64>>>(ds, mode);
    hipMemcpy(hs, ds, sizeof(hs), hipMemcpyDeviceToHost);
    printf("== ds_read_b64_tr_b16 mode %d (values are halfword indices in LDS)\n", mode);
    bad = 0;
    for (int l = 0; l < 64; ++l) {
      for (int e = 0; e < 4; ++e) {
        int want;
        if (mode == 0) want = (l & 15) + e * 16 + (l >> 4) * 64;
        else want = ((l >> 5) * 512 + e * 64 + (l & 31) * 2) / 2;   // row e, column lane&31 of the half's [8][32] tile
        if (hs[l * 4 + e] != want) bad++;
      }
      if (l < 20 || (l >= 32 && l < 36)) printf("  lane %2d: %4d %4d %4d %4d\n", l, hs[l * 4], hs[l * 4 + 1], hs[l * 4 + 2], hs[l * 4 + 3]);
    }
    printf("  expectation %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  }
  return 0;
}
