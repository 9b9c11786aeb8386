// Baseline JPEG marker parser + table-driven Huffman entropy decoder (host threads).
// See jpeg_entropy.h.  T.81 section / figure numbers are cited where they define the behaviour.
#include "jpeg_entropy.h"

#include <cstring>
#include <exception>

#include "../../include/lanefront.h"

namespace lf {
namespace jpeg {

namespace {

// zigzag index -> natural index (T.81 figure A.6)
const uint8_t kZigzag[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};

// Typical Huffman tables of T.81 Annex K.3 (code length counts, then symbols), for streams that ship
// without DHT segments (motion-JPEG style camera frames).
const uint8_t kDcLum[16 + 12] = { 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
const uint8_t kDcChr[16 + 12] = { 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
const uint8_t kAcLum[16 + 162] = {
    0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d,
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91,
    0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a,
    0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53,
    0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79,
    0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9,
    0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa
};
const uint8_t kAcChr[16 + 162] = {
    0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77,
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14,
    0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17,
    0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a,
    0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78,
    0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7,
    0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa
};

constexpr int kLook = 9;      // lookahead bits of the fast table

struct HuffTable {
    bool present = false;
    uint16_t fast[1 << kLook];    // (length << 8) | symbol for codes of up to kLook bits, 0 otherwise
    int32_t maxcode[18];          // largest code of each length, -1 if none (index = length)
    int32_t delta[17];            // symbol index = code + delta[length]
    uint8_t vals[256];

    // T.81 Annex C: canonical codes from the per-length counts
    bool build(const uint8_t* counts, const uint8_t* symbols, int n_symbols)
    {
        std::memset(fast, 0, sizeof(fast));
        std::memcpy(vals, symbols, (size_t)n_symbols);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            const int cnt = counts[len - 1];
            delta[len] = k - code;
            if (code + cnt > (1 << len)) return false;            // over-subscribed
            for (int i = 0; i < cnt; ++i, ++k, ++code) {
                if (len <= kLook) {
                    const int first = code << (kLook - len);
                    for (int f = 0; f < (1 << (kLook - len)); ++f) fast[first + f] = (uint16_t)((len << 8) | symbols[k]);
                }
            }
            maxcode[len] = cnt ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
        return true;
    }
};

// MSB-first bit reader over the entropy-coded segment.  0xFF00 is a stuffed 0xFF; any other marker
// stops the real data (zero bits are fed from then on and counted, so that a decoder that runs into
// them can be detected: `n < fake` means bits that were never in the stream have been consumed).
struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int n = 0;          // valid bits in acc (the low n bits)
    int fake = 0;       // how many of them are padding
    bool stopped = false;

    inline void fill()
    {
        while (n <= 56) {
            uint32_t c = 0;
            if (!stopped) {
                if (p >= end) stopped = true;
                else if (*p != 0xFF) c = *p++;
                else if (p + 1 < end && p[1] == 0x00) { c = 0xFF; p += 2; }
                else stopped = true;                             // a marker (or a dangling 0xFF)
            }
            if (stopped) fake += 8;
            acc = (acc << 8) | c;
            n += 8;
        }
    }
    inline uint32_t peek(int k) const { return (uint32_t)(acc >> (n - k)) & ((1u << k) - 1u); }
    inline void skip(int k) { n -= k; }
    inline bool overrun() const { return n < fake; }
};

inline int decode_symbol(BitReader& br, const HuffTable& t)
{
    if (br.n < 16) br.fill();
    const uint32_t e = t.fast[br.peek(kLook)];
    if (e) { br.skip((int)(e >> 8)); return (int)(e & 255u); }
    const int32_t code16 = (int32_t)br.peek(16);
    for (int len = kLook + 1; len <= 16; ++len) {
        const int32_t c = code16 >> (16 - len);
        if (c <= t.maxcode[len]) { br.skip(len); return t.vals[(c + t.delta[len]) & 255]; }
    }
    return -1;
}

// T.81 F.2.2.1: s additional bits, sign-extended
inline int receive_extend(BitReader& br, int s)
{
    if (br.n < s) br.fill();
    const int v = (int)br.peek(s);
    br.skip(s);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

inline int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

struct Component { int id, h, v, tq, td, ta; };

struct Stream {
    int rows = 0, cols = 0, ncomp = 0;
    Component comp[3];
    uint16_t qt[4][64];
    bool qt_present[4] = { false, false, false, false };
    HuffTable dc[4], ac[4];
    int restart = 0;
    int hmax = 1, vmax = 1;
    bool is_rgb = false;
    const uint8_t* scan = nullptr;
    const uint8_t* end = nullptr;
};

// Everything up to and including the SOS header.  lf_status.
int parse_headers(const uint8_t* d, size_t size, Stream& j)
{
    if (d == nullptr || size < 4 || d[0] != 0xFF || d[1] != 0xD8) return LF_ERR_DECODE;
    size_t pos = 2;
    bool have_sof = false, saw_jfif = false, saw_adobe = false;
    int adobe_transform = 0;
    for (;;) {
        if (pos + 4 > size || d[pos] != 0xFF) return LF_ERR_DECODE;
        while (pos < size && d[pos] == 0xFF) ++pos;
        if (pos >= size) return LF_ERR_DECODE;
        const int m = d[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return LF_ERR_DECODE;
        if (pos + 2 > size) return LF_ERR_DECODE;
        const int len = rd16(d + pos);
        if (len < 2 || pos + (size_t)len > size) return LF_ERR_DECODE;
        const uint8_t* s = d + pos + 2;
        const int n = len - 2;
        switch (m) {
        case 0xDB:
            for (int k = 0; k < n;) {
                const int pq = s[k] >> 4, tq = s[k] & 15;
                ++k;
                if (tq > 3 || pq > 1 || k + 64 * (pq + 1) > n) return LF_ERR_DECODE;
                for (int i = 0; i < 64; ++i) j.qt[tq][kZigzag[i]] = (uint16_t)(pq ? rd16(s + k + 2 * i) : s[k + i]);
                j.qt_present[tq] = true;
                k += 64 * (pq + 1);
            }
            break;
        case 0xC4:
            for (int k = 0; k < n;) {
                if (k + 17 > n) return LF_ERR_DECODE;
                const int tc = s[k] >> 4, th = s[k] & 15;
                if (tc > 1 || th > 3) return LF_ERR_DECODE;
                int total = 0;
                for (int i = 0; i < 16; ++i) total += s[k + 1 + i];
                if (total > 256 || k + 17 + total > n) return LF_ERR_DECODE;
                if (!(tc ? j.ac[th] : j.dc[th]).build(s + k + 1, s + k + 17, total)) return LF_ERR_DECODE;
                k += 17 + total;
            }
            break;
        case 0xC0:
        case 0xC1:
            if (n < 6 || s[0] != 8) return LF_ERR_UNSUPPORTED;
            j.rows = rd16(s + 1);
            j.cols = rd16(s + 3);
            j.ncomp = s[5];
            if (j.rows <= 0 || j.cols <= 0) return LF_ERR_DECODE;
            if (j.ncomp != 1 && j.ncomp != 3) return LF_ERR_UNSUPPORTED;
            if (n < 6 + 3 * j.ncomp) return LF_ERR_DECODE;
            for (int c = 0; c < j.ncomp; ++c) {
                Component& cp = j.comp[c];
                cp.id = s[6 + 3 * c];
                cp.h = s[7 + 3 * c] >> 4;
                cp.v = s[7 + 3 * c] & 15;
                cp.tq = s[8 + 3 * c];
                if (cp.h < 1 || cp.h > 4 || cp.v < 1 || cp.v > 4 || cp.tq > 3) return LF_ERR_DECODE;
            }
            have_sof = true;
            break;
        case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            return LF_ERR_UNSUPPORTED;                 // progressive, lossless, differential, arithmetic
        case 0xDD:
            if (n < 2) return LF_ERR_DECODE;
            j.restart = rd16(s);
            break;
        case 0xE0:
            if (n >= 5 && std::memcmp(s, "JFIF", 5) == 0) saw_jfif = true;
            break;
        case 0xEE:
            if (n >= 12 && std::memcmp(s, "Adobe", 5) == 0) { saw_adobe = true; adobe_transform = s[11]; }
            break;
        case 0xDA: {
            if (!have_sof) return LF_ERR_DECODE;
            if (n < 1 || s[0] != j.ncomp) return LF_ERR_UNSUPPORTED;      // one interleaved scan only
            if (n < 1 + 2 * j.ncomp + 3) return LF_ERR_DECODE;
            for (int c = 0; c < j.ncomp; ++c) {
                if (s[1 + 2 * c] != j.comp[c].id) return LF_ERR_UNSUPPORTED;
                j.comp[c].td = s[2 + 2 * c] >> 4;
                j.comp[c].ta = s[2 + 2 * c] & 15;
                if (j.comp[c].td > 3 || j.comp[c].ta > 3) return LF_ERR_DECODE;
            }
            const uint8_t* t = s + 1 + 2 * j.ncomp;
            if (t[0] != 0 || t[1] != 63 || t[2] != 0) return LF_ERR_UNSUPPORTED;
            j.scan = d + pos + len;
            j.end = d + size;
            break;
        }
        default:
            break;                                     // APPn, COM, ...: skipped
        }
        if (j.scan) break;
        pos += (size_t)len;
    }
    if (j.ncomp == 3) {
        // libjpeg's colour-space guess: JFIF says YCbCr; an Adobe marker says what its transform flag
        // says; otherwise component ids 'R','G','B' mean RGB
        if (saw_jfif) j.is_rgb = false;
        else if (saw_adobe) j.is_rgb = adobe_transform == 0;
        else j.is_rgb = j.comp[0].id == 'R' && j.comp[1].id == 'G' && j.comp[2].id == 'B';
    }
    if (j.ncomp == 1) {
        j.comp[0].h = j.comp[0].v = 1;                 // a lone component is never subsampled
        j.hmax = j.vmax = 1;
    } else {
        j.hmax = j.comp[0].h;
        j.vmax = j.comp[0].v;
        if (j.comp[1].h != 1 || j.comp[1].v != 1 || j.comp[2].h != 1 || j.comp[2].v != 1) return LF_ERR_UNSUPPORTED;
        if (!((j.hmax == 1 && j.vmax == 1) || (j.hmax == 2 && j.vmax == 1) || (j.hmax == 2 && j.vmax == 2)))
            return LF_ERR_UNSUPPORTED;
    }
    for (int c = 0; c < j.ncomp; ++c) {
        if (!j.qt_present[j.comp[c].tq]) return LF_ERR_DECODE;
        if (!j.dc[j.comp[c].td].present || !j.ac[j.comp[c].ta].present) {
            if (!j.dc[0].present) j.dc[0].build(kDcLum, kDcLum + 16, 12);
            if (!j.dc[1].present) j.dc[1].build(kDcChr, kDcChr + 16, 12);
            if (!j.ac[0].present) j.ac[0].build(kAcLum, kAcLum + 16, 162);
            if (!j.ac[1].present) j.ac[1].build(kAcChr, kAcChr + 16, 162);
            if (!j.dc[j.comp[c].td].present || !j.ac[j.comp[c].ta].present) return LF_ERR_DECODE;
        }
    }
    return LF_OK;
}

}  // namespace

int peek(const uint8_t* data, size_t size, int* rows, int* cols, int* ncomp, int* hmax, int* vmax)
{
    Stream j;
    const int rc = parse_headers(data, size, j);
    if (rc != LF_OK) return rc;
    if (rows) *rows = j.rows;
    if (cols) *cols = j.cols;
    if (ncomp) *ncomp = j.ncomp;
    if (hmax) *hmax = j.hmax;
    if (vmax) *vmax = j.vmax;
    return LF_OK;
}

static int decode_coefficients_impl(const uint8_t* data, size_t size, FrameCoefs& out, int expect_rows, int expect_cols)
{
    out.n_entries = 0;
    out.hdr.valid = 0;
    out.hdr.nblocks = 0;
    out.rows = out.cols = 0;
    Stream j;
    int rc = parse_headers(data, size, j);
    if (rc != LF_OK) { out.status = rc; return rc; }
    out.rows = j.rows;
    out.cols = j.cols;
    // A stream of another size than the caller declared is refused HERE, before anything is sized from the
    // stream's own (untrusted) SOF fields.
    if (expect_rows > 0 && expect_cols > 0 && (j.rows != expect_rows || j.cols != expect_cols)) {
        out.status = LF_ERR_BAD_ARG;
        return out.status;
    }
    FrameHeader& h = out.hdr;
    h.ncomp = j.ncomp;
    h.hmax = j.hmax;
    h.vmax = j.vmax;
    h.mcux = (j.cols + 8 * j.hmax - 1) / (8 * j.hmax);
    h.mcuy = (j.rows + 8 * j.vmax - 1) / (8 * j.vmax);
    h.is_rgb = j.is_rgb ? 1 : 0;
    const int luma_blocks = j.ncomp == 1 ? 1 : j.hmax * j.vmax;
    const int bpm = j.ncomp == 1 ? 1 : luma_blocks + 2;
    const long nblocks = (long)h.mcux * h.mcuy * bpm;
    if (nblocks > (1L << 24)) { out.status = LF_ERR_UNSUPPORTED; return out.status; }
    // Every block costs at least two bits of entropy data (a DC code and an EOB or AC code of >= 1 bit each) and
    // every coefficient at least two (a code and >= 1 magnitude bit): a stream too short for its declared size
    // cannot decode, so nothing is allocated beyond what the stream's own length can fill.
    const size_t scan_len = (size_t)(j.end - j.scan);
    const size_t max_units = 4 * scan_len + 64;
    if ((size_t)nblocks > max_units) { out.status = LF_ERR_DECODE; return out.status; }
    h.nblocks = (int32_t)nblocks;
    for (int c = 0; c < 3; ++c)
        std::memcpy(h.qt[c], j.qt[j.comp[c < j.ncomp ? c : 0].tq], sizeof(h.qt[c]));
    if (out.block_end.size() < (size_t)nblocks) out.block_end.resize((size_t)nblocks);

    BitReader br;
    br.p = j.scan;
    br.end = j.end;
    int pred[3] = { 0, 0, 0 };
    int next_rst = 0;
    size_t ne = 0;
    uint32_t* ent = out.entries.data();
    size_t cap = out.entries.size();
    long b = 0;
    const long n_mcu = (long)h.mcux * h.mcuy;
    for (long mcu = 0; mcu < n_mcu; ++mcu) {
        if (j.restart && mcu > 0 && mcu % j.restart == 0) {
            // T.81 E.2.4: the interval ends with 0..7 padding bits, then RSTm
            br.fill();
            if (!br.stopped || br.n - br.fake >= 8 || br.p + 2 > br.end || br.p[0] != 0xFF || br.p[1] != 0xD0 + next_rst) {
                rc = LF_ERR_DECODE;
                break;
            }
            br.p += 2;
            br.acc = 0; br.n = 0; br.fake = 0; br.stopped = false;
            next_rst = (next_rst + 1) & 7;
            pred[0] = pred[1] = pred[2] = 0;
        }
        for (int r = 0; r < bpm; ++r, ++b) {
            const int c = r < luma_blocks ? 0 : r - luma_blocks + 1;
            const HuffTable& tdc = j.dc[j.comp[c].td];
            const HuffTable& tac = j.ac[j.comp[c].ta];
            if (ne + 64 > cap) {
                size_t want = cap ? cap * 2 : (size_t)nblocks * 8 + 64;
                if (want > max_units + 64) want = max_units + 64;
                if (want < ne + 64) { rc = LF_ERR_DECODE; break; }      // more coefficients than the stream has bits for
                cap = want;
                out.entries.resize(cap);
                ent = out.entries.data();
            }
            int s = decode_symbol(br, tdc);
            if (s < 0 || s > 11) { rc = LF_ERR_DECODE; break; }
            if (s) pred[c] = (int)((unsigned)pred[c] + (unsigned)receive_extend(br, s));   // wraps, never UB (hostile streams)
            if (pred[c]) ent[ne++] = (uint32_t)(uint16_t)(int16_t)pred[c];
            for (int k = 1; k < 64;) {
                const int rs = decode_symbol(br, tac);
                if (rs < 0) { rc = LF_ERR_DECODE; break; }
                const int run = rs >> 4, sz = rs & 15;
                if (sz == 0) {
                    if (run != 15) break;              // EOB
                    k += 16;                            // ZRL
                    continue;
                }
                k += run;
                if (k > 63) { rc = LF_ERR_DECODE; break; }
                const int v = receive_extend(br, sz);
                ent[ne++] = ((uint32_t)kZigzag[k] << 16) | (uint32_t)(uint16_t)(int16_t)v;
                ++k;
            }
            if (rc != LF_OK || br.overrun()) { rc = LF_ERR_DECODE; break; }
            out.block_end[(size_t)b] = (uint32_t)ne;
        }
        if (rc != LF_OK) break;
    }
    if (rc != LF_OK) {
        out.status = rc;
        h.nblocks = 0;
        return rc;
    }
    out.n_entries = ne;
    h.valid = 1;
    out.status = LF_OK;
    return LF_OK;
}

int prepare_device_frame(const uint8_t* data, size_t size, int expect_rows, int expect_cols, DevFrame& out, size_t* scan_begin)
{
    out.hdr.valid = 0;
    out.hdr.nblocks = 0;
    out.scan_off = out.scan_len = 0;
    if (scan_begin) *scan_begin = 0;
    Stream j;
    int rc = parse_headers(data, size, j);
    if (rc != LF_OK) return rc;
    if (expect_rows > 0 && expect_cols > 0 && (j.rows != expect_rows || j.cols != expect_cols)) return LF_ERR_BAD_ARG;
    FrameHeader& h = out.hdr;
    h.ncomp = j.ncomp;
    h.hmax = j.hmax;
    h.vmax = j.vmax;
    h.mcux = (j.cols + 8 * j.hmax - 1) / (8 * j.hmax);
    h.mcuy = (j.rows + 8 * j.vmax - 1) / (8 * j.vmax);
    h.is_rgb = j.is_rgb ? 1 : 0;
    const int luma_blocks = j.ncomp == 1 ? 1 : j.hmax * j.vmax;
    const int bpm = j.ncomp == 1 ? 1 : luma_blocks + 2;
    const long nblocks = (long)h.mcux * h.mcuy * bpm;
    if (nblocks > (1L << 24)) return LF_ERR_UNSUPPORTED;
    const size_t scan_len = (size_t)(j.end - j.scan);
    if ((size_t)nblocks > 4 * scan_len + 64) return LF_ERR_DECODE;          // same bound as the host decoder: too short for its size
    if (scan_len >= (1u << 28)) return LF_ERR_UNSUPPORTED;
    h.nblocks = (int32_t)nblocks;
    h.entry_base = h.block_base = 0;
    for (int c = 0; c < 3; ++c)
        std::memcpy(h.qt[c], j.qt[j.comp[c < j.ncomp ? c : 0].tq], sizeof(h.qt[c]));
    out.restart = j.restart;
    out.n_mcu = h.mcux * h.mcuy;
    out.bpm = bpm;
    out.luma = luma_blocks;
    for (int c = 0; c < 3; ++c) {
        const Component& cp = j.comp[c < j.ncomp ? c : 0];
        out.tab_dc[c] = cp.td;
        out.tab_ac[c] = 4 + cp.ta;
    }
    for (int t = 0; t < 8; ++t) {
        const HuffTable& src = t < 4 ? j.dc[t] : j.ac[t - 4];
        HuffDev& d = out.tabs[t];
        d.present = src.present ? 1 : 0;
        if (!src.present) { std::memset(&d, 0, sizeof(d)); continue; }
        std::memcpy(d.fast, src.fast, sizeof(d.fast));
        std::memcpy(d.maxcode, src.maxcode, sizeof(d.maxcode));
        std::memcpy(d.delta, src.delta, sizeof(d.delta));
        std::memcpy(d.vals, src.vals, sizeof(d.vals));
    }
    out.scan_len = (uint32_t)scan_len;
    if (scan_begin) *scan_begin = (size_t)(j.scan - data);
    h.valid = 1;
    return LF_OK;
}

int decode_coefficients(const uint8_t* data, size_t size, FrameCoefs& out, int expect_rows, int expect_cols)
{
    // never let an exception cross into a worker thread or the C ABI
    try {
        return decode_coefficients_impl(data, size, out, expect_rows, expect_cols);
    } catch (const std::exception&) {
        out.n_entries = 0;
        out.hdr.valid = 0;
        out.hdr.nblocks = 0;
        out.status = LF_ERR_CAPACITY;      // out of host memory
        return out.status;
    }
}

WorkerPool::~WorkerPool()
{
    {
        std::lock_guard<std::mutex> lk(m_);
        stop_ = true;
    }
    start_.notify_all();
    for (std::thread& t : threads_) t.join();
}

void WorkerPool::loop(int id)
{
    unsigned long seen = 0;
    for (;;) {
        const std::function<void(int)>* job = nullptr;
        {
            std::unique_lock<std::mutex> lk(m_);
            start_.wait(lk, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            if (id < active_) job = job_;
        }
        if (job) {
            (*job)(id);
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_one();
        }
    }
}

void WorkerPool::run(int n_workers, const std::function<void(int)>& job)
{
    if (n_workers <= 1) { job(0); return; }
    while ((int)threads_.size() < n_workers) {
        const int id = (int)threads_.size();
        threads_.emplace_back([this, id] { loop(id); });
    }
    std::unique_lock<std::mutex> lk(m_);
    job_ = &job;
    active_ = n_workers;
    pending_ = n_workers;
    ++generation_;
    start_.notify_all();
    done_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
}

}  // namespace jpeg
}  // namespace lf
