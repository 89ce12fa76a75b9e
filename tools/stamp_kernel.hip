// Diagnostic (tools/exp_graph_timeline.py): a one-thread kernel that appends (label, wall clock)
// to a device buffer -- enqueued between the launches of a captured playout graph, it gives
// the timeline of the replay (s_memrealtime: 100 MHz).
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void stamp_kernel(uint64_t *buf, int *idx, int capacity, int label)
{
    const int i = atomicAdd(idx, 1);
    if (i < capacity) {
        buf[2 * i] = (uint64_t)label;
        buf[2 * i + 1] = wall_clock64();
    }
}

extern "C" int stamp(uint64_t *buf, int *idx, int capacity, int label, void *stream)
{
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, buf, idx, capacity, label);
    return (int)hipGetLastError();
}
