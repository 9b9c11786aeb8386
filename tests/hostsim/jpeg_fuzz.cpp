// Test harness only (CPU, AddressSanitizer + UBSan): the product's host-side JPEG entropy decoder
// (lane_slam_amd/csrc/jpeg_entropy.cpp) and the oracle's decoder (oracle/lf_oracle_jpeg.c) on hostile input --
// every prefix length class, random bit flips, random byte splices of the golden streams.  Memory errors abort
// under the sanitizers; when both decoders accept a mutated stream their coefficients must agree.
//
//   jpeg_fuzz <file.jpg>... [--iters N]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../lane_slam_amd/csrc/jpeg_entropy.cpp"

extern "C" {
int lfo_jpeg_info(const uint8_t* data, size_t size, int* rows, int* cols, int* ncomp, int* hmax, int* vmax);
int lfo_jpeg_coefficients(const uint8_t* data, size_t size, int16_t* qcoef, int cap_blocks, int* n_blocks);
int lfo_jpeg_decode(const uint8_t* data, size_t size, uint8_t* bgr);
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return rng_state; }

static long checked = 0, both_ok = 0;

static void one(const std::vector<uint8_t>& d)
{
    lf::jpeg::FrameCoefs fc;
    const int rc = lf::jpeg::decode_coefficients(d.data(), d.size(), fc);
    int rows = 0, cols = 0, nc = 0, hm = 0, vm = 0;
    const int irc = lfo_jpeg_info(d.data(), d.size(), &rows, &cols, &nc, &hm, &vm);
    ++checked;
    if (irc != 0) return;
    if ((long)rows * cols > 4096L * 4096L) return;                       // a mutated header may announce a huge image
    const int mcux = (cols + 8 * hm - 1) / (8 * hm), mcuy = (rows + 8 * vm - 1) / (8 * vm);
    const long nb = (long)mcux * mcuy * (nc == 1 ? 1 : hm * vm + 2);
    if (nb > (1L << 20)) return;
    std::vector<int16_t> ref((size_t)nb * 64);
    int got_blocks = 0;
    const int orc = lfo_jpeg_coefficients(d.data(), d.size(), ref.data(), (int)nb, &got_blocks);
    std::vector<uint8_t> px((size_t)rows * cols * 3);
    (void)lfo_jpeg_decode(d.data(), d.size(), px.data());
    if (rc == 0 && orc == 0) {
        ++both_ok;
        if (fc.hdr.nblocks != nb) { fprintf(stderr, "block count differs\n"); exit(1); }
        for (long b = 0; b < nb; ++b) {
            int16_t dense[64] = { 0 };
            const uint32_t e0 = b ? fc.block_end[(size_t)b - 1] : 0u, e1 = fc.block_end[(size_t)b];
            for (uint32_t e = e0; e < e1; ++e) dense[fc.entries[e] >> 16] = (int16_t)(fc.entries[e] & 0xffffu);
            if (memcmp(dense, &ref[(size_t)b * 64], sizeof(dense)) != 0) { fprintf(stderr, "coefficients differ in block %ld\n", b); exit(1); }
        }
    }
}

int main(int argc, char** argv)
{
    int iters = 300;
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--iters") && i + 1 < argc) { iters = atoi(argv[++i]); continue; }
        FILE* f = fopen(argv[i], "rb");
        if (!f) { perror(argv[i]); return 2; }
        std::vector<uint8_t> d;
        uint8_t buf[4096];
        size_t n;
        while ((n = fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + n);
        fclose(f);
        seeds.push_back(d);
    }
    for (const auto& s : seeds) {
        one(s);
        for (size_t cut = 0; cut < s.size(); cut += 1 + s.size() / 97) one(std::vector<uint8_t>(s.begin(), s.begin() + cut));
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> m = s;
            const int kind = rnd() % 4;
            if (kind == 0) m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8));
            else if (kind == 1) { for (int k = 0; k < 4; ++k) m[rnd() % m.size()] = (uint8_t)rnd(); }
            else if (kind == 2) { const size_t a = rnd() % m.size(), len = rnd() % 64; for (size_t k = a; k < m.size() && k < a + len; ++k) m[k] = 0xFF; }
            else {
                const size_t a = rnd() % m.size();
                size_t len = rnd() % 32;
                if (len > m.size() - a) len = m.size() - a;
                m.erase(m.begin() + a, m.begin() + a + len);
            }
            if (!m.empty()) one(m);
        }
    }
    printf("jpeg_fuzz: %ld inputs, %ld accepted by both decoders with identical coefficients\n", checked, both_ok);
    return 0;
}
