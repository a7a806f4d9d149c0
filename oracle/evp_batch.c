/* evp_batch.c -- CPU baseline of BASELINE config 5 (bench.py --config cfg5): n independent packets, each with its OWN
 * key and IV, through the system libcrypto's EVP_aes_*_gcm -- one EVP_EncryptInit_ex(key, iv) per packet, which is what
 * a per-packet key costs a CPU library (key schedule + GHASH table setup per packet; the reference's per-frame key
 * reload is tb/gcm_gctr.py:144-175).
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (lives under oracle/; never linked into or loaded by the product library).
 * The loop is in C so that the baseline measures libcrypto, not Python call overhead (five ctypes calls per 4 KiB
 * packet would cost more than the encryption).  libcrypto stands in for pycryptodome, which is not installed
 * (BASELINE.md section 2). */
#include <openssl/evp.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define API __attribute__((visibility("default")))
#else
#define API
#endif

/* packets [0, n): keys + p*key_len, ivs + p*12 (96-bit IV), pt/ct + p*pkt_len, tags + p*16.  Returns 0, or -1 on any EVP failure. */
API int evp_batch_encrypt(size_t n, size_t key_len, const uint8_t *keys, const uint8_t *ivs, const uint8_t *pt, size_t pkt_len,
                          uint8_t *ct, uint8_t *tags) {
    const EVP_CIPHER *ciph = key_len == 16 ? EVP_aes_128_gcm() : key_len == 24 ? EVP_aes_192_gcm() : key_len == 32 ? EVP_aes_256_gcm() : NULL;
    if (!ciph) return -1;
    EVP_CIPHER_CTX *c = EVP_CIPHER_CTX_new();
    if (!c) return -1;
    int rc = 0, outl = 0;
    if (EVP_EncryptInit_ex(c, ciph, NULL, NULL, NULL) != 1 || EVP_CIPHER_CTX_ctrl(c, EVP_CTRL_GCM_SET_IVLEN, 12, NULL) != 1) rc = -1;
    for (size_t p = 0; p < n && !rc; p++) {
        if (EVP_EncryptInit_ex(c, NULL, NULL, keys + p * key_len, ivs + p * 12) != 1) { rc = -1; break; }
        if (pkt_len && EVP_EncryptUpdate(c, ct + p * pkt_len, &outl, pt + p * pkt_len, (int)pkt_len) != 1) { rc = -1; break; }
        if (EVP_EncryptFinal_ex(c, ct + p * pkt_len, &outl) != 1) { rc = -1; break; }
        if (EVP_CIPHER_CTX_ctrl(c, EVP_CTRL_GCM_GET_TAG, 16, tags + p * 16) != 1) { rc = -1; break; }
    }
    EVP_CIPHER_CTX_free(c);
    return rc;
}

/* Frames under ONE key (round 6): n messages delimited by offset arrays (n + 1 uint64 entries each, as aesgcm_packets_crypt_dev takes them), one
 * EVP_*Init_ex(iv) per frame -- the key schedule and the GHASH table stay, as the RTL keeps H while no new key is loaded (src/gcm_gctr.vhd:142-144) and takes a new IV
 * per frame (src/aes_icb.vhd:60-70).  Frame p: data bytes [data_off[p], data_off[p + 1]) - data_base of in / out, AAD bytes [aad_off[p], aad_off[p + 1]) - aad_base of
 * aad (aad_off NULL: no AAD), so that a CHUNK of a large call can be checked with buffers that hold only that chunk.  Encrypt only.  Two users: bench.py --config frames
 * (cpu_baseline: what a CPU library does for such traffic) and the GPU tests of mixed calls, which check every one of 2^18 messages against libcrypto. */
API int evp_frames_crypt(size_t n, size_t key_len, const uint8_t *key, const uint8_t *ivs, const uint8_t *aad, const uint64_t *aad_off, uint64_t aad_base,
                         const uint8_t *in, const uint64_t *data_off, uint64_t data_base, uint8_t *out, uint8_t *tags) {
    const EVP_CIPHER *ciph = key_len == 16 ? EVP_aes_128_gcm() : key_len == 24 ? EVP_aes_192_gcm() : key_len == 32 ? EVP_aes_256_gcm() : NULL;
    if (!ciph) return -1;
    EVP_CIPHER_CTX *c = EVP_CIPHER_CTX_new();
    if (!c) return -1;
    int rc = 0, outl = 0;
    uint8_t scratch[16];
    if (EVP_EncryptInit_ex(c, ciph, NULL, NULL, NULL) != 1 || EVP_CIPHER_CTX_ctrl(c, EVP_CTRL_GCM_SET_IVLEN, 12, NULL) != 1 ||
        EVP_EncryptInit_ex(c, NULL, NULL, key, NULL) != 1) rc = -1;
    for (size_t p = 0; p < n && !rc; p++) {
        const uint64_t d0 = data_off[p] - data_base, len = data_off[p + 1] - data_off[p];
        const uint64_t a0 = aad_off ? aad_off[p] - aad_base : 0, alen = aad_off ? aad_off[p + 1] - aad_off[p] : 0;
        if (len > 0x7FFFFFFF || alen > 0x7FFFFFFF) { rc = -2; break; }
        if (EVP_EncryptInit_ex(c, NULL, NULL, NULL, ivs + p * 12) != 1) { rc = -1; break; }
        if (alen && EVP_EncryptUpdate(c, NULL, &outl, aad + a0, (int)alen) != 1) { rc = -1; break; }
        if (len && EVP_EncryptUpdate(c, out + d0, &outl, in + d0, (int)len) != 1) { rc = -1; break; }
        if (EVP_EncryptFinal_ex(c, scratch, &outl) != 1) { rc = -1; break; }
        if (EVP_CIPHER_CTX_ctrl(c, EVP_CTRL_GCM_GET_TAG, 16, tags + p * 16) != 1) { rc = -1; break; }
    }
    EVP_CIPHER_CTX_free(c);
    return rc;
}
