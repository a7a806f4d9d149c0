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
