// pack_host.cpp -- host-side repack of molecule calls into unique (variant, barcode) calls.
//
// Replaces the O(M log M) numpy work of Demultiplexer.pack_calls in the reference
// (demuxalot/demux.py:332-365: per-chromosome np.searchsorted on a structured array;
// :276-283: np.unique on a 12-byte structured dtype + np.multiply.at), which dominates the
// reference's end-to-end predict_posteriors (~3.5 us per molecule call).
//
// Own design, not a transliteration: 64-bit integer keys, one binary search per call against
// the sorted variant keys, and a stable LSD radix sort of (variant, barcode) keys so that the
// float32 products are taken in the input order of the members, which is what
// np.multiply.at does (sequential, starting from 1.0f).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "dmx_internal.h"

namespace {

inline uint64_t variant_key(int32_t chrom, int32_t pos, uint8_t base)
{
    return (uint64_t(uint32_t(chrom)) << 35) | (uint64_t(uint32_t(pos)) << 3) | uint64_t(base & 7);
}

// Stable LSD radix sort of idx[] by key[] (both permuted), 16-bit digits over the used bits.
void radix_sort_pairs(std::vector<uint64_t> &key, std::vector<uint32_t> &idx, uint64_t max_key)
{
    const size_t n = key.size();
    if (n < 2) return;
    std::vector<uint64_t> key2(n);
    std::vector<uint32_t> idx2(n);
    int bits = 0;
    while (bits < 64 && (max_key >> bits) != 0) bits++;
    std::vector<size_t> hist(1 << 16);
    for (int shift = 0; shift < bits; shift += 16) {
        std::fill(hist.begin(), hist.end(), 0);
        for (size_t i = 0; i < n; i++) hist[(key[i] >> shift) & 0xFFFF]++;
        size_t run = 0;
        for (size_t d = 0; d < hist.size(); d++) {
            size_t c = hist[d];
            hist[d] = run;
            run += c;
        }
        for (size_t i = 0; i < n; i++) {
            size_t dst = hist[(key[i] >> shift) & 0xFFFF]++;
            key2[dst] = key[i];
            idx2[dst] = idx[i];
        }
        key.swap(key2);
        idx.swap(idx2);
    }
}

}  // namespace

extern "C" int dmx_pack_calls_host(int64_t n_variants, const int32_t *var_chrom, const int32_t *var_pos,
                                   const uint8_t *var_base, int64_t n_calls, const int32_t *call_chrom,
                                   const int32_t *call_pos, const uint8_t *call_base, const int32_t *call_cb,
                                   const float *call_p, int32_t *call_variant, int64_t *n_matched,
                                   int64_t *n_unique, int32_t *out_variant, int32_t *out_cb, float *out_p, int64_t *out_count,
                                   int64_t *mol_per_variant)
{
    if (n_variants < 0 || n_calls < 0 || !n_matched || !n_unique)
        return dmx::fail(DMX_ERR_INVALID, "dmx_pack_calls_host: bad sizes or null counters");
    if (n_calls >= (int64_t(1) << 32))
        return dmx::fail(DMX_ERR_UNSUPPORTED, "dmx_pack_calls_host: more than 2^32 molecule calls in one batch");
    if (n_calls > 0 && (!call_chrom || !call_pos || !call_base || !call_cb || !call_p || !out_variant || !out_cb ||
                        !out_p || !out_count))
        return dmx::fail(DMX_ERR_INVALID, "dmx_pack_calls_host: null call arrays");
    if (n_variants > 0 && (!var_chrom || !var_pos || !var_base))
        return dmx::fail(DMX_ERR_INVALID, "dmx_pack_calls_host: null variant arrays");

    // 1. sorted variant keys -> row
    std::vector<std::pair<uint64_t, int32_t>> vkeys(n_variants);
    for (int64_t i = 0; i < n_variants; i++) vkeys[i] = {variant_key(var_chrom[i], var_pos[i], var_base[i]), int32_t(i)};
    std::sort(vkeys.begin(), vkeys.end());
    if (mol_per_variant) std::memset(mol_per_variant, 0, sizeof(int64_t) * n_variants);

    // 2. match every call; keep the matched ones in input order
    std::vector<uint64_t> key;
    std::vector<uint32_t> idx;
    key.reserve(n_calls);
    idx.reserve(n_calls);
    uint64_t max_key = 0;
    for (int64_t i = 0; i < n_calls; i++) {
        const uint64_t q = variant_key(call_chrom[i], call_pos[i], call_base[i]);
        auto it = std::lower_bound(vkeys.begin(), vkeys.end(), std::make_pair(q, int32_t(INT32_MIN)));
        const bool hit = it != vkeys.end() && it->first == q;
        if (call_variant) call_variant[i] = hit ? it->second : -1;
        if (!hit) continue;
        if (call_cb[i] < 0) return dmx::fail(DMX_ERR_INVALID, "dmx_pack_calls_host: negative barcode index");
        const uint64_t k = (uint64_t(uint32_t(it->second)) << 32) | uint32_t(call_cb[i]);
        key.push_back(k);
        idx.push_back(uint32_t(i));
        max_key = std::max(max_key, k);
        if (mol_per_variant) mol_per_variant[it->second]++;
    }
    *n_matched = int64_t(key.size());

    // 3. stable sort by (variant, barcode), then segment products in member order
    radix_sort_pairs(key, idx, max_key);
    int64_t u = -1;
    uint64_t prev = ~uint64_t(0);
    for (size_t i = 0; i < key.size(); i++) {
        if (i == 0 || key[i] != prev) {
            u++;
            prev = key[i];
            out_variant[u] = int32_t(key[i] >> 32);
            out_cb[u] = int32_t(key[i] & 0xFFFFFFFFu);
            out_p[u] = 1.0f;
            out_count[u] = 0;
        }
        out_p[u] = out_p[u] * call_p[idx[i]];  // float32, sequential: np.multiply.at semantics
        out_count[u]++;
    }
    *n_unique = u + 1;
    return DMX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// dmx_hash_host: a 64-bit content hash of a host buffer at memory bandwidth (several threads).
//
// The Python front-end keeps the packed problem resident on the device between calls on the same inputs
// (demuxalot_amd/demux.py: _pack_on_device).  The reference re-packs on every call (demux.py:303), so an array edited
// in place between two calls must be seen: the key of the resident problem is the hash of EVERY record of every
// container (round 5 sampled ~512 records: a sparse edit went unnoticed).  2 GB of records hash in ~15 ms on 16
// threads, against the 45 ms upload + 13 ms device pack a repack costs.
// The hash: per 1 MiB chunk four independent multiply-rotate lanes over 8-byte words (wyhash-style mixing, not
// cryptographic: it guards against edits, not adversaries), chunk hashes folded in chunk order with the length.
// ---------------------------------------------------------------------------------------------------------
#include <thread>

namespace {

inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t mix64(uint64_t h, uint64_t w)
{
    h ^= w * 0x9E3779B97F4A7C15ull;
    h = rotl64(h, 29) * 0xBF58476D1CE4E5B9ull;
    return h;
}

uint64_t hash_chunk(const uint8_t *p, size_t n, uint64_t seed)
{
    uint64_t h0 = seed ^ 0x243F6A8885A308D3ull, h1 = seed ^ 0x13198A2E03707344ull, h2 = seed ^ 0xA4093822299F31D0ull,
             h3 = seed ^ 0x082EFA98EC4E6C89ull;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        std::memcpy(w, p + i, 32);
        h0 = mix64(h0, w[0]);
        h1 = mix64(h1, w[1]);
        h2 = mix64(h2, w[2]);
        h3 = mix64(h3, w[3]);
    }
    uint64_t tail[4] = {0, 0, 0, 0};
    if (i < n) {
        std::memcpy(tail, p + i, n - i);
        h0 = mix64(h0, tail[0]);
        h1 = mix64(h1, tail[1]);
        h2 = mix64(h2, tail[2]);
        h3 = mix64(h3, tail[3]);
    }
    uint64_t h = mix64(mix64(mix64(mix64(uint64_t(n), h0), h1), h2), h3);
    h ^= h >> 31;
    return h * 0x94D049BB133111EBull;
}

}  // namespace

extern "C" int dmx_hash_host(const void *data, int64_t bytes, int32_t threads, uint64_t *hash_out)
{
    if (bytes < 0 || (bytes > 0 && data == nullptr) || hash_out == nullptr) return DMX_ERR_INVALID;
    const size_t chunk = size_t(1) << 20;
    const size_t n_chunks = (size_t(bytes) + chunk - 1) / chunk;
    std::vector<uint64_t> parts(n_chunks);
    const uint8_t *p = static_cast<const uint8_t *>(data);
    auto work = [&](size_t first, size_t step) {
        for (size_t c = first; c < n_chunks; c += step) {
            size_t off = c * chunk, len = std::min(chunk, size_t(bytes) - off);
            parts[c] = hash_chunk(p + off, len, uint64_t(c));
        }
    };
    int n_threads = threads > 0 ? threads : int(std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency())));
    n_threads = int(std::min<size_t>(size_t(n_threads), std::max<size_t>(1, n_chunks / 4)));
    if (n_threads <= 1) {
        work(0, 1);
    } else {
        std::vector<std::thread> pool;
        for (int t = 1; t < n_threads; t++) pool.emplace_back(work, size_t(t), size_t(n_threads));
        work(0, size_t(n_threads));
        for (auto &t : pool) t.join();
    }
    uint64_t h = mix64(0x452821E638D01377ull, uint64_t(bytes));
    for (size_t c = 0; c < n_chunks; c++) h = mix64(h, parts[c]);
    *hash_out = h ^ (h >> 32);
    return DMX_OK;
}
