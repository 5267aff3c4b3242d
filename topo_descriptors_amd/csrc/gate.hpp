// Device side of the ghost-row gate (common.hpp: struct Gate).
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace topo {

// `slot`: the block's byte of g.skipped.  Called by every thread of a block before the block's first read of a ghost row; true = the ghost rows are in
// place.  One thread polls the word the communication stream writes after the exchange (relaxed, system scope:
// the writer is another queue of this GPU), sleeping between polls so that the block's other waves keep their issue
// slots; the acquire fence behind the barrier (buffer_inv sc0 sc1) drops whatever this CU and its L2 hold of the
// rows the receive kernel has just written.  In the expected case - the seam parts come last, the exchange is long
// over - this is one load and one barrier per block.  A block that has waited `limit_ticks` gives up (false): it
// leaves its seam tiles to the clean-up launch behind the exchange's event (disc_wave_impl.hpp, launch_parts) and
// counts itself in `*timeouts`, a statistic (topo_amd_gate_giveups).  Nothing ever hangs on the exchange, and the
// bits do not depend on who computed a seam tile.  Lean mode (g.errors set, Context::gate_mode): there is no clean-up
// launch, the limit is long, and a block whose wait runs out counts itself in `*errors` and goes on - the next
// synchronising call of the library fails with that.
__device__ __forceinline__ bool gate_wait(const Gate& g, const unsigned slot) {
    if (g.word == nullptr) return true;
    uint8_t* mine = g.skipped + slot;  // the block's verdict travels through its own byte (no LDS of the kernel's is touched)
    if (threadIdx.x == 0) {
        const long long t0 = __builtin_amdgcn_s_memtime();
        uint8_t gave_up = 0;
        while ((int)(__hip_atomic_load(g.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - g.epoch) < 0) {
            if ((unsigned long long)(__builtin_amdgcn_s_memtime() - t0) > (unsigned long long)g.limit_ticks) {
                __hip_atomic_fetch_add(g.errors ? g.errors : g.timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                gave_up = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        __hip_atomic_store(mine, gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    const bool open = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    return open;
}

}  // namespace topo
