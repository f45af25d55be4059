// state_blob.h -- checkpoint / resume of a context's device state as one host blob (used by the masking, multiband and
// MVDR contexts; the main context has its own in api.hip).  A blob = header {magic, version, hash of the configuration,
// a few host-side counters} followed by the device buffers in a fixed order.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

namespace mca {

struct BlobPart { void *ptr; size_t bytes; };
struct BlobHeader { unsigned magic; int version; unsigned cfg_hash; int pad; long long host[4]; };

inline unsigned blob_fnv(const void *data, size_t n, unsigned h = 2166136261u)
{
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 16777619u; }
    return h;
}

inline long long blob_size(const std::vector<BlobPart> &parts)
{
    long long n = sizeof(BlobHeader);
    for (const BlobPart &p : parts) n += (long long)p.bytes;
    return n;
}

// 0 = ok, 1 = blob too small / NULL, 2 = HIP error
inline int blob_save(const std::vector<BlobPart> &parts, const BlobHeader &h, void *blob, long long bytes)
{
    if (!blob || bytes < blob_size(parts)) return 1;
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    unsigned char *out = static_cast<unsigned char *>(blob);
    std::memcpy(out, &h, sizeof(h)); out += sizeof(h);
    for (const BlobPart &p : parts) { if (hipMemcpy(out, p.ptr, p.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 2; out += p.bytes; }
    return 0;
}

// 0 = ok, 1 = NULL / truncated, 2 = HIP error, 3 = not a blob of this kind / version, 4 = other configuration
inline int blob_load(const std::vector<BlobPart> &parts, unsigned magic, unsigned cfg_hash, const void *blob, long long bytes, BlobHeader *h_out)
{
    if (!blob || bytes < (long long)sizeof(BlobHeader)) return 1;
    BlobHeader h;
    std::memcpy(&h, blob, sizeof(h));
    if (h.magic != magic || h.version != 1) return 3;
    if (h.cfg_hash != cfg_hash) return 4;
    if (bytes < blob_size(parts)) return 1;
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    const unsigned char *in = static_cast<const unsigned char *>(blob) + sizeof(h);
    for (const BlobPart &p : parts) { if (hipMemcpy(p.ptr, in, p.bytes, hipMemcpyHostToDevice) != hipSuccess) return 2; in += p.bytes; }
    *h_out = h;
    return 0;
}

inline const char *blob_error(int rc)
{
    switch (rc) {
    case 1: return "state blob is NULL, truncated or smaller than the state size";
    case 2: return "HIP error while copying the state";
    case 3: return "not a state blob of this module / library version";
    case 4: return "state blob was saved by a context with a different configuration";
    default: return "";
    }
}

}  // namespace mca
