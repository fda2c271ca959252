// device_index.cpp — the per-device copies of an index handle (rank blocks, contraction entries,
// two-base blocks) and the knobs that shape them.
#include "capi_internal.hpp"

namespace kbo_host {

int g_waves_per_cu = 0;
bool g_force_big = false;
uint64_t g_pair_min_rows = 24ull << 20;

int current_device()
{
    int dev = -1;
    HIP_OK(hipGetDevice(&dev));
    return dev;
}

kbo::DevIndexView device_view(kbo_index *idx, int device)
{
    std::lock_guard<std::mutex> g(idx->mu);
    auto it = idx->dev.find(device);
    if (it == idx->dev.end()) {
        KBO_REQUIRE(idx->host.n_sets < 0xFFFFFFF0ull, KBO_E_UNSUPPORTED,
                    "n_sets >= 2^32: 64-bit device layout not built yet");
        // two-base extension blocks: worth their 2.7 B/row once the one-base blocks stop fitting L2
        // (the walk is then bound by line fills, and a two-base step needs one instead of two)
        const size_t est_rank = (idx->host.n_sets / 96 + 2) * 64, est_ent = (idx->host.n_sets + 2) * 12;
        const bool want_pairs = idx->host.n_sets >= g_pair_min_rows && !g_force_big &&
                                est_rank * 5 + est_ent + 64 < 0xFFFFFFF0ull;
        kbo::DeviceLayout lay;
        kbo::make_device_layout(idx->host, lay, want_pairs);
        int prev = current_device();
        if (prev != device) HIP_OK(hipSetDevice(device));
        DevCopy *dc = new DevCopy();
        try {
            const size_t per = lay.n_blocks * 16;
            // arena = rank blocks of A,C,G,T | one all-zero "null" block | contraction entries.
            // When that exceeds the 32-bit offset range (n_sets * 12 B of entries >= ~4 GiB) the
            // entries get their own allocation and 64-bit offsets ("big" kernels).
            const size_t ent_bytes = lay.ent.size() * sizeof(uint32_t);
            const size_t rank_bytes = per * 4 + 16;
            KBO_REQUIRE(rank_bytes < 0xFFFFFFF0ull, KBO_E_UNSUPPORTED, "rank blocks >= 4 GiB");
            dc->big = g_force_big || rank_bytes + ent_bytes >= 0xFFFFFFF0ull;
            const size_t pair_bytes = dc->big ? 0 : lay.pair.size() * sizeof(uint32_t);
            const size_t base_bytes = ((dc->big ? rank_bytes : rank_bytes + ent_bytes) + 15) / 16 * 16;
            const size_t arena_bytes = base_bytes + pair_bytes;
            dc->arena.alloc(arena_bytes);
            HIP_OK(hipMemset(dc->arena.p, 0, arena_bytes));
            for (int c = 0; c < 4; c++)
                HIP_OK(hipMemcpy(dc->arena.as<uint8_t>() + per * c, lay.rank[c].data(), per,
                                 hipMemcpyHostToDevice));
            if (dc->big) {
                dc->ent.alloc(ent_bytes + 16);
                HIP_OK(hipMemcpy(dc->ent.p, lay.ent.data(), ent_bytes, hipMemcpyHostToDevice));
            } else {
                HIP_OK(hipMemcpy(dc->arena.as<uint8_t>() + rank_bytes, lay.ent.data(), ent_bytes,
                                 hipMemcpyHostToDevice));
            }
            if (pair_bytes) {
                HIP_OK(hipMemcpy(dc->arena.as<uint8_t>() + base_bytes, lay.pair.data(), pair_bytes, hipMemcpyHostToDevice));
                dc->pair_off = (uint32_t)(base_bytes / 16);
            }
            dc->n_blocks = lay.n_blocks;
            idx->rank_bytes = per * 4;
            idx->lcs_bytes = ent_bytes;
        } catch (...) {
            delete dc;
            if (prev != device) (void)hipSetDevice(prev);
            throw;
        }
        if (prev != device) HIP_OK(hipSetDevice(prev));
        it = idx->dev.emplace(device, dc).first;
    }
    DevCopy *dc = it->second;
    kbo::DevIndexView v;
    v.arena = dc->arena.as<uint4>();
    v.n_blocks = (uint32_t)dc->n_blocks;
    v.lcs_off = (uint32_t)(dc->n_blocks * 4 + 1);
    v.pair_off = dc->pair_off;
    v.ent = dc->big ? dc->ent.as<uint8_t>() : nullptr;
    v.big = dc->big ? 1u : 0u;
    v.n = (uint32_t)idx->host.n_sets;
    v.k = idx->host.k;
    return v;
}

// upper bound on resident walk waves: CUs x waves per CU (default 32 = 8 per SIMD)
int walk_max_waves()
{
    int dev = current_device();
    int cus = 0;
    HIP_OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int per = g_waves_per_cu > 0 ? g_waves_per_cu : 32;
    return std::max(1, cus) * per;
}

} // namespace kbo_host
