// Sharding of independent arrays over the GPUs of one node: rank r owns a contiguous block, the first n % world ranks one
// array more (the same blocks as mcarray_amd/dist.py: partition / local_range, which bench.py shards with).  Arrays never
// interact -- the reference keeps no state between module objects -- so nothing but the results is exchanged.
// No counterpart in the reference, whose driver (src/programs/mcabeamf.cpp:77-122) runs one stream on one thread.
#ifndef MCA_HIP_PARTITION_H
#define MCA_HIP_PARTITION_H
namespace mca {

struct ArrayBlock { int first, count; };

inline ArrayBlock localArrays(int nArrays, int rank, int world)
{
    const int base = nArrays / world, rem = nArrays % world;
    ArrayBlock b;
    b.count = base + (rank < rem ? 1 : 0);
    b.first = rank * base + (rank < rem ? rank : rem);
    return b;
}

}  // namespace mca
#endif
