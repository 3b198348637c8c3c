// openjph_amd/csrc/ojph_pool.h -- a small persistent host thread pool for the Tier-2 work around the
// kernels (packet-header coding, placement copies).  The reference is single-threaded; at GPU kernel
// speeds its serial Tier-2 (precinct::prepare_precinct / write, ojph_precinct.cpp:94-325) would be the
// whole frame time, and the units it codes -- the bands of a packet, the packets of a tile -- are
// independent of each other.
//
// parallel_for(n, fn) runs fn(0) .. fn(n-1) on the pool's threads and on the caller; several callers
// (the finisher threads of a frame pipeline) may be inside it at the same time.  OJPHGPU_T2_THREADS
// sets the number of threads working on one call (default min(hardware threads, 8); 1 = caller only).
// A body may throw: every item still runs (and is counted), and the first exception is rethrown from
// parallel_for on the calling thread once the last body has returned -- never on a pool thread.
#ifndef OJPH_POOL_H
#define OJPH_POOL_H

#include <cstddef>
#include <functional>

namespace ojphgpu {

unsigned pool_threads();
void parallel_for(size_t n, const std::function<void(size_t)>& fn);

}  // namespace ojphgpu

#endif
