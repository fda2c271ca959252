// synth.cpp — deterministic synthetic inputs for tests and bench.py:
// SplitMix64, counter-based (draw j = mix(seed + (j+1)*gamma)), base = "ACGT"[x >> 62].
// Bench/test tooling exported from the same shared library; not part of the kbo API.
#include <cstddef>
#include <cstdint>

namespace {
inline uint64_t draw(uint64_t seed, uint64_t j)
{
    uint64_t z = seed + (j + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline int code_of(uint8_t ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3; }
} // namespace

extern "C" {

// iid uniform genome of `len` bases
void kbo_synth_genome(uint64_t seed, uint8_t *out, uint64_t len)
{
    for (uint64_t i = 0; i < len; i++) out[i] = (uint8_t)"ACGT"[draw(seed, i) >> 62];
}

// `n_reads` forward-strand reads of `read_len` bases with uniform starts and per-base
// substitutions with probability sub_per_65536 / 65536; read r consumes draws
// [(first_read + r) * (read_len + 1), ...): one for the start, one per base.
void kbo_synth_reads(uint64_t seed, const uint8_t *genome, uint64_t genome_len, uint64_t first_read,
                     uint64_t n_reads, uint32_t read_len, uint32_t sub_per_65536, uint8_t *out)
{
    const uint64_t span = genome_len - read_len + 1;
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint64_t j0 = (first_read + r) * ((uint64_t)read_len + 1);
        const uint64_t start = draw(seed, j0) % span;
        uint8_t *o = out + r * read_len;
        for (uint32_t b = 0; b < read_len; b++) {
            const uint64_t x = draw(seed, j0 + 1 + b);
            int c = code_of(genome[start + b]);
            if ((x & 0xFFFF) < sub_per_65536) c = (c + 1 + (int)((x >> 16) % 3)) & 3;
            o[b] = (uint8_t)"ACGT"[c];
        }
    }
}

} // extern "C"
