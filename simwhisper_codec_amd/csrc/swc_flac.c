/* swc_flac: a self-contained FLAC decoder for the audio-input side of the path (the role torchaudio.load plays for
 * `*.flac` in the reference, utils/helpers.py:77-94, 105-111; LibriSpeech, the codec's evaluation corpus, is FLAC).
 * Host code, plain C, no dependency; built by simwhisper_codec_amd/build.py into libswc_io.so and bound with ctypes
 * (simwhisper_codec_amd/wavio.py).  Written from the format description (RFC 9639): STREAMINFO, frame header with CRC-8,
 * CONSTANT / VERBATIM / FIXED / LPC subframes, partitioned Rice residuals (4- and 5-bit parameters, escape partitions),
 * wasted bits, the three stereo decorrelation modes, frame CRC-16.
 * No other FLAC decoder exists in the build environment to compare with, so every decode VERIFIES ITSELF: both CRCs of
 * every frame are checked and the MD5 signature STREAMINFO carries is recomputed over the decoded samples; a file that
 * does not check out is an error, never silently wrong audio.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define FLAC_OK 0
#define FLAC_E_FORMAT (-1)   /* not a FLAC stream / malformed */
#define FLAC_E_CRC (-2)      /* a frame failed its CRC-8 or CRC-16 */
#define FLAC_E_UNSUP (-3)    /* reserved codes, > 8 channels, > 32 bits */
#define FLAC_E_SPACE (-4)    /* output buffer too small */
#define FLAC_E_MD5 (-5)      /* decoded audio does not match the stream's MD5 signature */

/* ---------------------------------------------------------------- bit reader (MSB first) */
typedef struct {
    const uint8_t* p;
    size_t n, pos;      /* pos: next byte */
    uint64_t acc;       /* left-aligned bit accumulator */
    int bits;           /* valid bits in acc */
    int err;
} br_t;

static void br_init(br_t* b, const uint8_t* p, size_t n, size_t start) {
    b->p = p; b->n = n; b->pos = start; b->acc = 0; b->bits = 0; b->err = 0;
}
static void br_fill(br_t* b) {
    while (b->bits <= 56 && b->pos < b->n) {
        b->acc |= (uint64_t)b->p[b->pos++] << (56 - b->bits);
        b->bits += 8;
    }
}
static uint32_t br_read(br_t* b, int k) { /* k <= 32 */
    if (k == 0) return 0;
    if (b->bits < k) br_fill(b);
    if (b->bits < k) { b->err = 1; return 0; }
    uint32_t v = (uint32_t)(b->acc >> (64 - k));
    b->acc <<= k;
    b->bits -= k;
    return v;
}
static int32_t br_read_signed(br_t* b, int k) {
    if (k == 0) return 0;
    uint32_t v = br_read(b, k);
    if (k < 32 && (v >> (k - 1))) v |= ~((1u << k) - 1u);
    return (int32_t)v;
}
static uint32_t br_unary(br_t* b) { /* number of 0 bits before the next 1 bit */
    uint32_t q = 0;
    for (;;) {
        if (b->bits == 0) br_fill(b);
        if (b->bits == 0) { b->err = 1; return 0; }
        if (b->acc == 0) { q += (uint32_t)b->bits; b->bits = 0; continue; }
        int z = __builtin_clzll(b->acc);
        if (z >= b->bits) { q += (uint32_t)b->bits; b->acc = 0; b->bits = 0; continue; }
        q += (uint32_t)z;
        b->acc = z == 63 ? 0 : b->acc << (z + 1);
        b->bits -= z + 1;
        return q;
    }
}
static size_t br_byte_pos(const br_t* b) { return b->pos - (size_t)(b->bits / 8); } /* only when byte aligned */
static void br_align(br_t* b) {
    int drop = b->bits & 7;
    b->acc <<= drop;
    b->bits -= drop;
}

/* ---------------------------------------------------------------- CRCs */
static uint8_t crc8(const uint8_t* p, size_t n) { /* poly x^8 + x^2 + x + 1, init 0 */
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= p[i];
        for (int k = 0; k < 8; ++k) c = (uint8_t)((c & 0x80) ? ((c << 1) ^ 0x07) : (c << 1));
    }
    return c;
}
static uint16_t crc16(const uint8_t* p, size_t n) { /* poly x^16 + x^15 + x^2 + 1, init 0 */
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= (uint16_t)((uint16_t)p[i] << 8);
        for (int k = 0; k < 8; ++k) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1));
    }
    return c;
}

/* ---------------------------------------------------------------- MD5 (RFC 1321) */
typedef struct { uint32_t h[4]; uint64_t len; uint8_t buf[64]; size_t fill; } md5_t;
static const uint32_t MD5_K[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
    0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
    0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
    0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
    0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
    0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
static const uint8_t MD5_S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20,
                                  5, 9, 14, 20, 5, 9, 14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                  6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
static void md5_block(md5_t* m, const uint8_t* p) {
    uint32_t w[16], a = m->h[0], b = m->h[1], c = m->h[2], d = m->h[3];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    for (int i = 0; i < 64; ++i) {
        uint32_t f; int g;
        if (i < 16) { f = (b & c) | (~b & d); g = i; }
        else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
        else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
        else { f = c ^ (b | ~d); g = (7 * i) & 15; }
        uint32_t t = a + f + MD5_K[i] + w[g];
        a = d; d = c; c = b;
        b = b + ((t << MD5_S[i]) | (t >> (32 - MD5_S[i])));
    }
    m->h[0] += a; m->h[1] += b; m->h[2] += c; m->h[3] += d;
}
static void md5_init(md5_t* m) {
    m->h[0] = 0x67452301; m->h[1] = 0xefcdab89; m->h[2] = 0x98badcfe; m->h[3] = 0x10325476; m->len = 0; m->fill = 0;
}
static void md5_update(md5_t* m, const uint8_t* p, size_t n) {
    m->len += n;
    while (n) {
        size_t k = 64 - m->fill; if (k > n) k = n;
        memcpy(m->buf + m->fill, p, k);
        m->fill += k; p += k; n -= k;
        if (m->fill == 64) { md5_block(m, m->buf); m->fill = 0; }
    }
}
static void md5_final(md5_t* m, uint8_t out[16]) {
    uint64_t bits = m->len * 8;
    uint8_t pad = 0x80;
    md5_update(m, &pad, 1);
    pad = 0;
    while (m->fill != 56) md5_update(m, &pad, 1);
    uint8_t l[8];
    for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (8 * i));
    md5_update(m, l, 8);
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) out[4 * i + k] = (uint8_t)(m->h[i] >> (8 * k));
}

/* ---------------------------------------------------------------- stream header */
typedef struct { int sr, ch, bps, min_bs, max_bs; int64_t total; uint8_t md5[16]; size_t first_frame; } info_t;

static int parse_header(const uint8_t* d, size_t n, info_t* s) {
    size_t pos = 0;
    if (n >= 10 && d[0] == 'I' && d[1] == 'D' && d[2] == '3') { /* skip an ID3v2 tag in front of the stream */
        size_t sz = ((size_t)(d[6] & 0x7f) << 21) | ((size_t)(d[7] & 0x7f) << 14) | ((size_t)(d[8] & 0x7f) << 7) | (size_t)(d[9] & 0x7f);
        pos = 10 + sz;
    }
    if (pos + 4 > n || memcmp(d + pos, "fLaC", 4) != 0) return FLAC_E_FORMAT;
    pos += 4;
    int have = 0;
    for (;;) {
        if (pos + 4 > n) return FLAC_E_FORMAT;
        int last = d[pos] >> 7, type = d[pos] & 0x7f;
        size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
        pos += 4;
        if (pos + len > n) return FLAC_E_FORMAT;
        if (type == 0) {
            if (len < 34) return FLAC_E_FORMAT;
            const uint8_t* q = d + pos;
            s->min_bs = (q[0] << 8) | q[1];
            s->max_bs = (q[2] << 8) | q[3];
            s->sr = (q[10] << 12) | (q[11] << 4) | (q[12] >> 4);
            s->ch = ((q[12] >> 1) & 7) + 1;
            s->bps = (((q[12] & 1) << 4) | (q[13] >> 4)) + 1;
            s->total = ((int64_t)(q[13] & 0xf) << 32) | ((int64_t)q[14] << 24) | ((int64_t)q[15] << 16) | ((int64_t)q[16] << 8) | q[17];
            memcpy(s->md5, q + 18, 16);
            have = 1;
        }
        pos += len;
        if (last) break;
    }
    if (!have || s->sr == 0 || s->bps < 4 || s->bps > 32) return FLAC_E_FORMAT;
    s->first_frame = pos;
    return FLAC_OK;
}

/* stream parameters without decoding: returns FLAC_OK and fills sr / channels / bits / total samples per channel
 * (0 = unknown: decode into a buffer sized by swc_flac_max_samples) */
int swc_flac_info(const uint8_t* data, size_t n, int32_t* sr, int32_t* channels, int32_t* bits, int64_t* total) {
    info_t s;
    int rc = parse_header(data, n, &s);
    if (rc != FLAC_OK) return rc;
    *sr = s.sr; *channels = s.ch; *bits = s.bps; *total = s.total;
    return FLAC_OK;
}

/* ---------------------------------------------------------------- residual + subframes */
static int read_residual(br_t* b, int64_t* out, int bs, int order) {
    int method = (int)br_read(b, 2);
    if (method > 1) return FLAC_E_UNSUP;
    int pbits = method ? 5 : 4, esc = method ? 31 : 15;
    int porder = (int)br_read(b, 4);
    int parts = 1 << porder;
    if ((bs % parts) != 0 || (bs >> porder) < order) return FLAC_E_FORMAT;
    int idx = order;
    for (int p = 0; p < parts; ++p) {
        int cnt = (bs >> porder) - (p == 0 ? order : 0);
        int k = (int)br_read(b, pbits);
        if (k == esc) {
            int raw = (int)br_read(b, 5);
            for (int i = 0; i < cnt; ++i) out[idx++] = br_read_signed(b, raw);
        } else {
            for (int i = 0; i < cnt; ++i) {
                uint64_t q = br_unary(b);
                uint64_t u = (q << k) | br_read(b, k);   /* 32-bit audio: folded residuals need more than 32 bits */
                out[idx++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
            }
        }
        if (b->err) return FLAC_E_FORMAT;
    }
    return FLAC_OK;
}

static int read_subframe(br_t* b, int64_t* s, int bs, int bps) {
    if (br_read(b, 1)) return FLAC_E_FORMAT; /* padding bit */
    int type = (int)br_read(b, 6);
    int wasted = 0;
    if (br_read(b, 1)) {
        const uint32_t z = br_unary(b); /* a crafted stream may carry more than 2^31 zero bits: compare before the cast */
        if (z >= (uint32_t)bps - 1u) return FLAC_E_FORMAT;
        wasted = (int)z + 1;
    }
    bps -= wasted;
    if (bps <= 0) return FLAC_E_FORMAT;
    int rc = FLAC_OK;
    if (type == 0) { /* CONSTANT */
        int64_t v = bps <= 32 ? (int64_t)br_read_signed(b, bps) : 0;
        if (bps > 32) { uint32_t hi = br_read(b, bps - 32), lo = br_read(b, 32); v = ((int64_t)hi << 32) | lo; if (hi >> (bps - 33)) v |= -((int64_t)1 << bps); }
        for (int i = 0; i < bs; ++i) s[i] = v;
    } else if (type == 1) { /* VERBATIM */
        for (int i = 0; i < bs; ++i) {
            if (bps <= 32) s[i] = br_read_signed(b, bps);
            else { uint32_t hi = br_read(b, bps - 32), lo = br_read(b, 32); int64_t v = ((int64_t)hi << 32) | lo; if (hi >> (bps - 33)) v |= -((int64_t)1 << bps); s[i] = v; }
        }
    } else if (type >= 8 && type <= 12) { /* FIXED, order type - 8 */
        int order = type - 8;
        if (order > bs) return FLAC_E_FORMAT;
        for (int i = 0; i < order; ++i) {
            if (bps <= 32) s[i] = br_read_signed(b, bps);
            else { uint32_t hi = br_read(b, bps - 32), lo = br_read(b, 32); int64_t v = ((int64_t)hi << 32) | lo; if (hi >> (bps - 33)) v |= -((int64_t)1 << bps); s[i] = v; }
        }
        rc = read_residual(b, s, bs, order);
        if (rc != FLAC_OK) return rc;
        for (int i = order; i < bs; ++i) {
            int64_t p = 0;
            switch (order) {
                case 1: p = s[i - 1]; break;
                case 2: p = 2 * s[i - 1] - s[i - 2]; break;
                case 3: p = 3 * s[i - 1] - 3 * s[i - 2] + s[i - 3]; break;
                case 4: p = 4 * s[i - 1] - 6 * s[i - 2] + 4 * s[i - 3] - s[i - 4]; break;
                default: break;
            }
            s[i] = (int64_t)((uint64_t)s[i] + (uint64_t)p);
        }
    } else if (type >= 32) { /* LPC, order (type & 31) + 1 */
        int order = (type & 31) + 1;
        if (order > bs) return FLAC_E_FORMAT;
        for (int i = 0; i < order; ++i) {
            if (bps <= 32) s[i] = br_read_signed(b, bps);
            else { uint32_t hi = br_read(b, bps - 32), lo = br_read(b, 32); int64_t v = ((int64_t)hi << 32) | lo; if (hi >> (bps - 33)) v |= -((int64_t)1 << bps); s[i] = v; }
        }
        int prec = (int)br_read(b, 4) + 1;
        if (prec == 16) return FLAC_E_UNSUP; /* 1111 is invalid */
        int shift = br_read_signed(b, 5);
        if (shift < 0) return FLAC_E_UNSUP;
        int32_t coef[32];
        for (int i = 0; i < order; ++i) coef[i] = br_read_signed(b, prec);
        rc = read_residual(b, s, bs, order);
        if (rc != FLAC_OK) return rc;
        for (int i = order; i < bs; ++i) {
            uint64_t p = 0;   /* wraps on damaged streams instead of overflowing; exact on valid ones (|sum| < 2^53) */
            for (int j = 0; j < order; ++j) p += (uint64_t)(int64_t)coef[j] * (uint64_t)s[i - 1 - j];
            s[i] = (int64_t)((uint64_t)s[i] + (uint64_t)((int64_t)p >> shift));
        }
    } else {
        return FLAC_E_UNSUP; /* reserved subframe types */
    }
    if (b->err) return FLAC_E_FORMAT;
    if (wasted) for (int i = 0; i < bs; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
    return FLAC_OK;
}

static const int BS_TABLE[16] = {0, 192, 576, 1152, 2304, 4608, -8, -16, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
static const int SR_TABLE[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
static const int BPS_TABLE[8] = {0, 8, 12, -1, 16, 20, 24, 32};

/* Decodes the whole stream.  out: interleaved int32 samples [frame][channel], cap = capacity in samples per channel.
 * Returns the number of samples per channel, or a negative FLAC_E_*.  *md5_state: 1 = signature verified, 0 = the stream
 * carries no signature (all zero); a mismatch is FLAC_E_MD5. */
int64_t swc_flac_decode(const uint8_t* data, size_t n, int32_t* out, int64_t cap, int32_t* md5_state) {
    info_t st;
    int rc = parse_header(data, n, &st);
    if (rc != FLAC_OK) return rc;
    if (st.ch > 8) return FLAC_E_UNSUP;
    const int maxbs = 65536;
    int64_t* chan = (int64_t*)malloc(sizeof(int64_t) * (size_t)maxbs * (size_t)st.ch);
    uint8_t* pcm = (uint8_t*)malloc((size_t)maxbs * (size_t)st.ch * 4);
    if (!chan || !pcm) { free(chan); free(pcm); return FLAC_E_SPACE; }
    md5_t md; md5_init(&md);
    const int bytes_ps = (st.bps + 7) / 8;
    int64_t done = 0;
    size_t pos = st.first_frame;
    rc = FLAC_OK;
    while (pos + 2 <= n) {
        if (!(data[pos] == 0xFF && (data[pos + 1] & 0xFE) == 0xF8)) {
            /* trailing bytes that are not a frame (e.g. an ID3v1 tag): stop once the announced length is reached */
            if (st.total && done >= st.total) break;
            rc = FLAC_E_FORMAT; break;
        }
        br_t b; br_init(&b, data, n, pos);
        br_read(&b, 15);                       /* sync + reserved */
        br_read(&b, 1);                        /* blocking strategy */
        int bs_code = (int)br_read(&b, 4), sr_code = (int)br_read(&b, 4);
        int ca = (int)br_read(&b, 4), ss_code = (int)br_read(&b, 3);
        if (br_read(&b, 1)) { rc = FLAC_E_FORMAT; break; }
        /* UTF-8-like coded frame / sample number (up to 36 bits): only its length matters here */
        uint32_t lead = br_read(&b, 8);
        int ones = 0;                          /* leading 1 bits of the first byte: 0 (one byte), or 2..7 (that many bytes) */
        while (ones < 8 && (lead & (0x80u >> ones))) ++ones;
        if (ones == 1 || ones == 8) { rc = FLAC_E_FORMAT; break; }
        for (int i = 1; i < ones; ++i) if ((br_read(&b, 8) & 0xC0) != 0x80) rc = FLAC_E_FORMAT;
        if (rc != FLAC_OK) break;
        int bs = BS_TABLE[bs_code];
        if (bs == 0) { rc = FLAC_E_UNSUP; break; }
        if (bs == -8) bs = (int)br_read(&b, 8) + 1;
        else if (bs == -16) bs = (int)br_read(&b, 16) + 1;
        if (sr_code == 12) br_read(&b, 8);
        else if (sr_code == 13 || sr_code == 14) br_read(&b, 16);
        else if (sr_code == 15) { rc = FLAC_E_UNSUP; break; }
        (void)SR_TABLE;
        size_t hdr_end = br_byte_pos(&b);
        uint32_t c8 = br_read(&b, 8);
        if (b.err) { rc = FLAC_E_FORMAT; break; }
        if (crc8(data + pos, hdr_end - pos) != (uint8_t)c8) { rc = FLAC_E_CRC; break; }
        int bps = BPS_TABLE[ss_code];
        if (bps == 0) bps = st.bps;
        if (bps < 0) { rc = FLAC_E_UNSUP; break; }
        int nch = ca < 8 ? ca + 1 : 2;
        if (ca > 10 || nch != st.ch || bps != st.bps || bs > maxbs) { rc = FLAC_E_UNSUP; break; }
        for (int c = 0; c < nch && rc == FLAC_OK; ++c) {
            int side = (ca == 8 && c == 1) || (ca == 9 && c == 0) || (ca == 10 && c == 1);
            rc = read_subframe(&b, chan + (size_t)c * maxbs, bs, bps + (side ? 1 : 0));
        }
        if (rc != FLAC_OK) break;
        br_align(&b);
        size_t body_end = br_byte_pos(&b);
        uint32_t c16 = br_read(&b, 16);
        if (b.err) { rc = FLAC_E_FORMAT; break; }
        if (crc16(data + pos, body_end - pos) != (uint16_t)c16) { rc = FLAC_E_CRC; break; }
        pos = body_end + 2;
        int64_t* a = chan; int64_t* s2 = chan + maxbs;
        if (ca == 8) { for (int i = 0; i < bs; ++i) s2[i] = a[i] - s2[i]; }                 /* left, side -> right = left - side */
        else if (ca == 9) { for (int i = 0; i < bs; ++i) a[i] = a[i] + s2[i]; }             /* side, right -> left = side + right */
        else if (ca == 10) {                                                                 /* mid, side */
            for (int i = 0; i < bs; ++i) {
                int64_t m = a[i], sd = s2[i];
                m = m * 2 + (sd & 1);
                a[i] = (m + sd) >> 1;
                s2[i] = (m - sd) >> 1;
            }
        }
        if (done + bs > cap) { rc = FLAC_E_SPACE; break; }
        size_t w = 0;
        for (int i = 0; i < bs; ++i)
            for (int c = 0; c < nch; ++c) {
                int64_t v = chan[(size_t)c * maxbs + i];
                out[(done + i) * nch + c] = (int32_t)v;
                for (int k = 0; k < bytes_ps; ++k) pcm[w++] = (uint8_t)((uint64_t)v >> (8 * k));
            }
        md5_update(&md, pcm, w);
        done += bs;
    }
    free(chan); free(pcm);
    if (rc != FLAC_OK) return rc;
    if (st.total && done != st.total) return FLAC_E_FORMAT;
    int zero = 1;
    for (int i = 0; i < 16; ++i) zero = zero && st.md5[i] == 0;
    if (zero) { *md5_state = 0; return done; }
    uint8_t dig[16];
    md5_final(&md, dig);
    if (memcmp(dig, st.md5, 16) != 0) return FLAC_E_MD5;
    *md5_state = 1;
    return done;
}

/* upper bound of samples per channel for a stream whose STREAMINFO does not say (total = 0): one 16-bit sample cannot take
 * less than ... nothing safe exists in general, so the caller grows the buffer on FLAC_E_SPACE; this helper just offers
 * a first guess from the file size (verbatim 8-bit mono would be n samples) */
int64_t swc_flac_max_samples(size_t n) { return (int64_t)n * 16 + 65536; }
