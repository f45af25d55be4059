// stage.h -- grow-only device staging buffers for the host-pointer entry points (*_frames_host): a streaming
// caller hands over one small chunk after another, and a hipMalloc/hipFree pair per buffer and call costs
// more than the kernels of a small chunk.
#pragma once
#include <hip/hip_runtime.h>

namespace mca {

struct StagePool {
    static constexpr int N = 8;
    void *p[N] = {};
    size_t cap[N] = {};
    // returns a device buffer of at least `bytes` bytes for slot i (nullptr on allocation failure or bytes == 0)
    void *get(int i, size_t bytes)
    {
        if (bytes == 0) return nullptr;
        if (bytes > cap[i]) {
            if (p[i]) (void)hipFree(p[i]);
            p[i] = nullptr; cap[i] = 0;
            const size_t want = bytes + bytes / 4;          // head room: chunk sizes of a stream vary a little
            if (hipMalloc(&p[i], want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            cap[i] = want;
        }
        return p[i];
    }
    void release()
    {
        for (int i = 0; i < N; ++i) { if (p[i]) (void)hipFree(p[i]); p[i] = nullptr; cap[i] = 0; }
    }
};

}  // namespace mca
