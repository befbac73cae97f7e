// Runtime of the host emulation (see hip_emu.h).  TEST INFRASTRUCTURE ONLY.
#include "hip_emu.h"

namespace emu {
thread_local Block* g_block = nullptr;
thread_local dim3 g_tid, g_bid, g_bdim, g_gdim;
thread_local unsigned g_coll = 0;

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    const unsigned nthreads = block.x * block.y * block.z;
    if (block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) {
        std::fprintf(stderr, "emu: only 1-D launches are supported\n");
        std::abort();
    }
    const unsigned nwaves = (nthreads + 63) / 64;
    for (unsigned b = 0; b < grid.x; ++b) {
        Block blk;
        blk.nthreads = nthreads;
        blk.block_barrier = std::make_unique<YieldBarrier>(nthreads);
        for (unsigned w = 0; w < nwaves; ++w) {
            unsigned lanes = std::min(64u, nthreads - w * 64);
            blk.wave_barrier.push_back(std::make_unique<YieldBarrier>(lanes));
        }
        blk.xa.assign(nwaves * 2 * 64, 0.f);           // (x 2: see emu::slot)
        blk.xb.assign(nwaves * 2 * 64, 0.f);
        blk.xw.assign(nwaves * 2 * 64 * 8, 0u);
        blk.smem.assign(smem + 64, 0x7f);   // poison: uninitialised LDS reads show up as NaN-ish garbage
        std::vector<std::thread> threads;
        threads.reserve(nthreads);
        for (unsigned t = 0; t < nthreads; ++t) {
            threads.emplace_back([&, t]() {
                g_block = &blk;
                g_coll = 0;
                g_tid = dim3(t);
                g_bid = dim3(b);
                g_bdim = block;
                g_gdim = grid;
                body();
            });
        }
        for (auto& th : threads) th.join();
    }
}
}  // namespace emu
