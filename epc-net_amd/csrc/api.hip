// Error reporting + version for libepcnet_hip.so.
#include <stdarg.h>
#include "common.h"
#include <cstring>

static thread_local char g_err[512] = "";

void epc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* epc_last_error(void) { return g_err; }
extern "C" int epc_version(void) { return 100; }

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8 on the host: the checksum TensorFlow's bundle
// format stores per tensor and per table block (tensorflow/core/lib/hash/crc32c.h).  Checkpoint payloads are tens of
// MB, which a Python loop cannot checksum in reasonable time.  crc = epc_crc32c(0, data, n); chainable.
static uint32_t g_crc_tab[8][256];
static bool g_crc_ready = false;
static void crc_init() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
        g_crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xff];
    g_crc_ready = true;
}
extern "C" uint32_t epc_crc32c(uint32_t crc, const void* data, size_t n) {
    if (!g_crc_ready) crc_init();  // idempotent: a race only repeats identical writes
    const unsigned char* p = (const unsigned char*)data;
    uint32_t c = ~crc;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = g_crc_tab[7][lo & 0xff] ^ g_crc_tab[6][(lo >> 8) & 0xff] ^ g_crc_tab[5][(lo >> 16) & 0xff] ^ g_crc_tab[4][lo >> 24] ^
            g_crc_tab[3][hi & 0xff] ^ g_crc_tab[2][(hi >> 8) & 0xff] ^ g_crc_tab[1][(hi >> 16) & 0xff] ^ g_crc_tab[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ g_crc_tab[0][(c ^ *p++) & 0xff];
    return ~c;
}
