// Microbenchmark (round 2): CTR-only AES, BITSLICED over 32 blocks per lane, S-box as a circuit of
// v_bitop3_b32 (tests/host_emul/aesgcm_bs.h), no LDS, no GHASH.  Writes the keystream to HBM (16 B per block, coalesced) like
// k_main<NR, MODE_KS> does, verifies sampled blocks against the literal byte-wise cipher on the host, and prints GB/s
// of keystream and the shader clock it sustained -- to be put beside k_main<NR, KS> (profiles/ks_time.py).
//   hipcc --offload-arch=gfx950 -O3 -o bs_ctr bs_ctr.hip && ./bs_ctr
#include "../../tests/host_emul/aesgcm_bs.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

struct BsParams {
    u32 iv0, iv1, iv2;   // IV, memory-order words
    u32 ctr0;            // counter of block 0
    u32 n_tiles;         // tiles of 2048 blocks (32 per lane x 64 lanes)
    unsigned char *out;
    unsigned long long *cyc;
};

// bit planes of the 32 counters c_i = A + 64 i (i = 0..31) of one lane: plane k bit i = bit k of c_i
__device__ __forceinline__ void counter_planes(u32 A, u32 *P) {
#pragma unroll
    for (int k = 0; k < 6; k++) P[k] = 0u - ((A >> k) & 1u);
    const u32 B = A >> 6;
    const u32 I[5] = {0xAAAAAAAAu, 0xCCCCCCCCu, 0xF0F0F0F0u, 0xFF00FF00u, 0xFFFF0000u};
    u32 carry = 0;
#pragma unroll
    for (int k = 0; k < 26; k++) {
        const u32 bk = 0u - ((B >> k) & 1u);
        if (k < 5) {
            P[6 + k] = BS_XOR3(bk, I[k], carry);
            carry = BS_LUT(bk, I[k], carry, 0xE8);                 // majority
        } else {
            P[6 + k] = bk ^ carry;
            carry = bk & carry;
        }
    }
}

template <int NR, int WPS>
__global__ __launch_bounds__(256, WPS) void k_bs_ctr(const u32 *__restrict__ rkm, const BsParams p) {   // rkm: (nr + 1) x 128 round-key bit masks, read through the scalar cache
    const u32 lane = threadIdx.x & 63u;
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const unsigned long long t0 = clock64(), wc0 = wall_clock64();
    for (u32 tile = wave; tile < p.n_tiles; tile += n_waves) {
        u32 s[128];
        // round 0: counter block IV || cnt (aes_icb.vhd:97-100,118) xor round key 0
        const u32 ivw[3] = {p.iv0, p.iv1, p.iv2};
#pragma unroll
        for (int j = 0; j < 12; j++)
#pragma unroll
            for (int b = 0; b < 8; b++) s[8 * j + b] = (0u - ((ivw[j >> 2] >> (8 * (j & 3) + b)) & 1u)) ^ rkm[8 * j + b];
        u32 P[32];
        counter_planes(p.ctr0 + tile * 2048u + lane, P);
#pragma unroll
        for (int j = 12; j < 16; j++)
#pragma unroll
            for (int b = 0; b < 8; b++) s[8 * j + b] = P[8 * (15 - j) + b] ^ rkm[8 * j + b];
        bs_rounds(s, rkm, NR);
        // back to one block per (lane, i): dword d of block i = transposed plane group d
        u32 w0[32], w1[32], w2[32], w3[32];
#pragma unroll
        for (int q = 0; q < 32; q++) { w0[q] = s[q]; w1[q] = s[32 + q]; w2[q] = s[64 + q]; w3[q] = s[96 + q]; }
        bs_transpose32(w0); bs_transpose32(w1); bs_transpose32(w2); bs_transpose32(w3);
        unsigned char *dst = p.out + ((size_t)tile * 2048u + lane) * 16u;
#pragma unroll
        for (int i = 0; i < 32; i++) gstore16(dst + (size_t)i * 1024u, make_uint4(w0[i], w1[i], w2[i], w3[i]));
    }
    const unsigned long long t1 = clock64(), wc1 = wall_clock64();
    if (lane == 0 && p.cyc) { p.cyc[2 * wave] = t1 - t0; p.cyc[2 * wave + 1] = wc1 - wc0; }
}

template <int NR, int WPS>
static void run(const char *name, int key_len, int n_cu, size_t bytes) {
    uint8_t sbox[256];
    for (u32 x = 0; x < 256; x++) sbox[x] = (uint8_t)sbox_calc(x);
    uint8_t key[32], rk[240], iv[12];
    for (int i = 0; i < 32; i++) key[i] = (uint8_t)(i * 7 + 3);
    for (int i = 0; i < 12; i++) iv[i] = (uint8_t)(0xA0 + i);
    const int nr = key_expand_bytes(key, key_len, sbox, rk);
    std::vector<u32> rkm(128 * (nr + 1));
    bs_key_masks(rk, nr, rkm.data());
    BsParams p;
    u32 *d_rkm; unsigned long long *d_cyc;
    hipMalloc(&d_rkm, rkm.size() * 4); hipMemcpy(d_rkm, rkm.data(), rkm.size() * 4, hipMemcpyHostToDevice);
    const int wgs = n_cu * WPS * 4 / 4;                       // WPS waves per SIMD: 4 x WPS waves per CU = WPS workgroups of 256
    hipMalloc(&d_cyc, 16 * (size_t)wgs * 4);
    hipMalloc(&p.out, bytes);
    p.cyc = d_cyc;
    p.iv0 = load_le32(iv); p.iv1 = load_le32(iv + 4); p.iv2 = load_le32(iv + 8);
    p.ctr0 = 2; p.n_tiles = (u32)(bytes / (2048 * 16));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_bs_ctr<NR, WPS>), dim3(wgs), dim3(256), 0, 0, d_rkm, p);      // warm-up
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_bs_ctr<NR, WPS>), dim3(wgs), dim3(256), 0, 0, d_rkm, p);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    std::vector<unsigned long long> h(2 * (size_t)wgs * 4);
    hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double cs = 0, ws = 0; for (size_t i = 0; i < h.size() / 2; i++) { cs += (double)h[2 * i]; ws += (double)h[2 * i + 1]; }
    // verify sampled blocks against the literal cipher
    int bad = 0;
    const size_t n_blocks = bytes / 16;
    const size_t samples[] = {0, 1, 61, 62, 63, 64, 2047, 2048, 2049, 4095, n_blocks / 2 + 77, n_blocks - 2049, n_blocks - 1};
    for (size_t bi : samples) {
        uint8_t got[16], want[16], ctr[16];
        hipMemcpy(got, p.out + 16 * bi, 16, hipMemcpyDeviceToHost);
        memcpy(ctr, iv, 12);
        const u32 c = 2 + (u32)bi;
        ctr[12] = (uint8_t)(c >> 24); ctr[13] = (uint8_t)(c >> 16); ctr[14] = (uint8_t)(c >> 8); ctr[15] = (uint8_t)c;
        aes_block_bytes(rk, nr, sbox, ctr, want);
        if (memcmp(got, want, 16)) bad++;
    }
    printf("%-40s %8.3f ms  %8.1f GB/s keystream  (%7.1f GiB/s)   sclk %4.0f MHz   verify %s  %s\n", name, best, bytes / (best * 1e-3) / 1e9,
           bytes / (best * 1e-3) / (double)(1ull << 30), ws > 0 ? cs / ws * 100.0 : 0.0, bad ? "MISMATCH" : "ok", err == hipSuccess ? "" : hipGetErrorString(err));
    hipFree(d_rkm); hipFree(d_cyc); hipFree(p.out);
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 4096) << 20;
    printf("%s, %d CUs, %zu MiB of keystream per launch, bitsliced AES-CTR (S-box = 90 v_bitop3/v_xor per 32 S-boxes)\n", prop.name, n_cu, bytes >> 20);
    run<10, 2>("AES-128 CTR bitsliced, 2 waves/SIMD", 16, n_cu, bytes);
    run<10, 1>("AES-128 CTR bitsliced, 1 wave/SIMD", 16, n_cu, bytes);
    run<14, 2>("AES-256 CTR bitsliced, 2 waves/SIMD", 32, n_cu, bytes);
    run<14, 1>("AES-256 CTR bitsliced, 1 wave/SIMD", 32, n_cu, bytes);
    return 0;
}
