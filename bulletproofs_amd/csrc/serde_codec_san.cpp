// CPU-only harness around the serde wire-format codec (serde_codec.cpp), built with -fsanitize=address,undefined by
// `make san` and run by tests/test_host_sanitizers.py.  The codec parses untrusted input (JSON text / bincode bytes), so
// it is driven with round trips of synthetic canonical proofs, every truncation of their encodings and 20 000
// single-byte corruptions per format; every call must return (BP_OK or BP_EINVAL) without a sanitizer report, and an
// accepted corruption must re-encode to a canonical proof.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "serde_codec.cpp"

namespace bp {
void set_error_msg(const char*) {}
}  // namespace bp

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

static std::vector<uint8_t> synth(int kind, size_t k) {
    const size_t n = (kind == 0 ? 292 : 195) + 66 * k;
    std::vector<uint8_t> p(n);
    for (auto& b : p) b = (uint8_t)rnd();
    auto pt = [&](size_t off) { p[off] = 2 + (rnd() & 1); };
    if (kind == 0) {
        for (int i = 0; i < 4; i++) pt(33 * i);
        for (size_t i = 0; i < 2 * k; i++) pt(228 + 33 * i);
        if (k & 1) memset(&p[n - 64], 0, 40);  // a short BigInt
        if (k == 2) memset(&p[n - 32], 0, 32);  // a zero BigInt
    } else {
        pt(0);
        for (size_t i = 0; i < 2 * k + 2; i++) pt(33 + 33 * i);
    }
    return p;
}

int main() {
    int fails = 0;
    for (int kind = 0; kind < 2; kind++)
        for (size_t k : {0, 1, 2, 6, 12}) {
            std::vector<uint8_t> proof = synth(kind, k);
            for (int fmt = 0; fmt < 2; fmt++) {
                size_t need = 0;
                bp_proof_to_serde(kind, fmt, proof.data(), proof.size(), nullptr, 0, &need);
                std::vector<uint8_t> enc(need);
                if (bp_proof_to_serde(kind, fmt, proof.data(), proof.size(), enc.data(), enc.size(), &need) != BP_OK) fails++;
                std::vector<uint8_t> back(proof.size() + 8);
                size_t bl = 0;
                if (bp_proof_from_serde(kind, fmt, enc.data(), enc.size(), back.data(), back.size(), &bl) != BP_OK || bl != proof.size() ||
                    memcmp(back.data(), proof.data(), bl))
                    fails++;
                // every truncation (heap copies of the exact length, so that an over-read is a report)
                for (size_t cut = 0; cut < enc.size(); cut += (enc.size() > 4000 ? 7 : 1)) {
                    std::vector<uint8_t> t(enc.begin(), enc.begin() + cut);
                    size_t l2 = 0;
                    if (bp_proof_from_serde(kind, fmt, t.data(), t.size(), back.data(), back.size(), &l2) == BP_OK) fails++;
                }
                // single-byte corruptions
                for (int it = 0; it < 2000; it++) {
                    std::vector<uint8_t> t(enc);
                    t[rnd() % t.size()] = (uint8_t)rnd();
                    size_t l2 = 0;
                    if (bp_proof_from_serde(kind, fmt, t.data(), t.size(), back.data(), back.size(), &l2) == BP_OK) {
                        size_t n3 = 0;
                        std::vector<uint8_t> re(enc.size() + 4096);
                        if (bp_proof_to_serde(kind, fmt, back.data(), l2, re.data(), re.size(), &n3) != BP_OK) fails++;
                    }
                }
            }
        }
    // garbage
    for (int it = 0; it < 2000; it++) {
        std::vector<uint8_t> t(rnd() % 300);
        for (auto& b : t) b = "{}[]\",:0123456789abcdefLRA_ \\"[rnd() % 30];
        uint8_t out[64];
        size_t l2 = 0;
        if (bp_proof_from_serde(it & 1, 0, t.data(), t.size(), out, sizeof out, &l2) == BP_OK) fails++;
    }
    std::string deep(100000, '[');
    uint8_t out[64];
    size_t l2 = 0;
    if (bp_proof_from_serde(0, 0, (const uint8_t*)deep.data(), deep.size(), out, sizeof out, &l2) == BP_OK) fails++;
    if (fails) {
        fprintf(stderr, "FAIL: %d\n", fails);
        return 1;
    }
    printf("serde_codec_san OK\n");
    return 0;
}
