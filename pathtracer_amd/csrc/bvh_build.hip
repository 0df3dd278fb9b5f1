// BVH::Build (src/bvh.cpp:204-228, SplitMethod::SAH) on the GPU: the same tree, node for node, as the reference's
// recursive builder - same split axis, same 12-bucket costs, same element order after every std::partition and
// std::nth_element call - built breadth first, one level of the recursion per round of kernels.
//
// What makes that possible:
//   * every leaf holds one shape (a range of more than four shapes is always split, src/bvh.cpp:95-99,149-158, and
//     smaller ones are halved by nth_element), so the subtree over n shapes has 2n - 1 nodes and a node's slot in
//     the pre-order array is known as soon as its range is: first child = slot + 1, second child = slot + 2 * (shapes
//     in the first child) (src/bvh.cpp:180-202);
//   * boxes are unions (min / max), bucket counts are integer sums: no result depends on the order of the reduction
//     (but for the sign of a zero bound: the reference keeps the first it meets, the reduction here the lesser /
//     greater in IEEE total order; nothing computed from a box tells the two apart);
//   * std::partition over a range is a fixed permutation: the k-th element from the left that fails the predicate and
//     lies before the partition point changes places with the k-th element from the right that passes it and lies
//     behind - positions a prefix sum of the predicate gives to every element at once;
//   * the bucket sums of a range (count, box, centroid box per bucket) hold, as unions over the buckets on either
//     side of the chosen split, the boxes and centroid boxes of its two halves: one pass over the shapes per level
//     feeds both the split of this level and the axes of the next;
//   * ranges of at most 16 shapes are finished by one thread each, which walks the reference's recursion itself, with
//     the element moves of libstdc++'s std::partition and std::nth_element (GCC 11; on at most four elements: one
//     median-of-three partition step for four, then insertion sort).
// All arithmetic is float32 in the reference's operation order, compiled without contraction.
//
// Per level: BucketKernel (shapes) -> SplitKernel (ranges) -> FlagKernel, three scan kernels, PartnerKernel,
// SwapKernel (shapes); SubtreeKernel for the small ranges the level before set aside; one 12-byte read-back tells
// the host how many ranges the next level has.
#include "pt_api.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdio>

namespace ptd
{
int SetError( int code, const char* message ); // api.hip: the message pt_last_error returns
}

namespace
{

constexpr uint32_t kNone       = 0xFFFFFFFFu;
constexpr int kBuckets         = 12; // src/bvh.cpp:102
constexpr uint32_t kSmallRange = 4;  // src/bvh.cpp:95,150
constexpr uint32_t kThreadRange = 16; // a range of at most this many shapes is finished by one thread
constexpr int kBlock           = 1024;
constexpr int kWaves           = kBlock / 64;
constexpr int kMaxRuns         = 64; // ranges a block may meet and still combine their sums in LDS
constexpr float kFltMax        = 3.402823466e+38f; // the empty box: min = +this, max = -this (src/aabb.cpp:6-10)
constexpr uint32_t kMaxLevels  = 8192;

// float <-> unsigned with the same order, so that integer atomic min / max reduce floats
__host__ __device__ __forceinline__ uint32_t Enc( float f )
{
    uint32_t b;
    __builtin_memcpy( &b, &f, 4 );
    return b ^ ( ( b >> 31 ) ? 0xFFFFFFFFu : 0x80000000u );
}
__host__ __device__ __forceinline__ float Dec( uint32_t u )
{
    const uint32_t b = u ^ ( ( u >> 31 ) ? 0x80000000u : 0xFFFFFFFFu );
    float f;
    __builtin_memcpy( &f, &b, 4 );
    return f;
}

// glm::min / glm::max (glm/detail/func_common.inl:16-20)
__device__ __forceinline__ float Gmin( float x, float y ) { return ( y < x ) ? y : x; }
__device__ __forceinline__ float Gmax( float x, float y ) { return ( x < y ) ? y : x; }

struct Box
{
    float lo[3], hi[3];
};

__device__ __forceinline__ Box EmptyBox() { return Box{ { kFltMax, kFltMax, kFltMax }, { -kFltMax, -kFltMax, -kFltMax } }; }

__device__ __forceinline__ void Union( Box& a, const Box& b ) // src/aabb.cpp:18-22
{
    for ( int k = 0; k < 3; ++k )
    {
        a.lo[k] = Gmin( a.lo[k], b.lo[k] );
        a.hi[k] = Gmax( a.hi[k], b.hi[k] );
    }
}

__device__ __forceinline__ float SurfaceArea( const Box& b ) // src/aabb.cpp:52-56
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2 * ( dx * dy + dx * dz + dy * dz );
}

__device__ __forceinline__ int LongestDimension( const Box& b ) // src/aabb.cpp:35-50
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if ( dx > dy && dx > dz ) return 0;
    return dy > dz ? 1 : 2;
}

// bucket of a centroid along the split axis (src/bvh.cpp:108-111 with AABB::Offset, src/aabb.cpp:58-66)
__device__ __forceinline__ int BucketOf( float centroid, float cMin, float cMax )
{
    float rel = centroid - cMin;
    if ( cMax > cMin ) rel /= cMax - cMin;
    const int b = static_cast< int >( kBuckets * rel );
    return b < kBuckets - 1 ? b : kBuckets - 1;
}

// One shape: {lo.xyz, shape index} {hi.xyz, -}.  The centroid, 0.5 * (lo + hi) (src/aabb.cpp:30-33), is recomputed
// where it is needed.
__device__ __forceinline__ float Centroid( float lo, float hi ) { return 0.5f * ( lo + hi ); }

// A range of more than kThreadRange shapes that this level splits.
struct BigRange
{
    uint32_t start, end, slot, cutoff;
    uint32_t left, right; // the ranges of the next level that the two halves become, or kNone
    uint32_t dim, noMove;
    uint32_t acc[12]; // Enc of: box lo xyz, centroid box lo xyz (minima), box hi xyz, centroid box hi xyz (maxima)
    float cMin, cMax; // centroid box along dim
    float area;       // SurfaceArea of the range's box
    uint32_t needReduce; // box and centroid box still have to be summed from the shapes
};
static_assert( sizeof( BigRange ) == 96, "BigRange is 24 words" );

struct SmallRange
{
    uint32_t start, end, slot;
};

// bucket sums of one range: [bucket][0] = count, [1..3] = Enc( box lo ), [4..6] = Enc( centroid box lo ) (minima),
// [7..9] = Enc( box hi ), [10..12] = Enc( centroid box hi ) (maxima)
constexpr uint32_t kBucketWords = 16;
struct Buckets
{
    uint32_t w[kBuckets][kBucketWords];
};

__device__ __forceinline__ void WriteNode( uint32_t* nodes, uint32_t slot, const Box& box, uint32_t offset, uint32_t numShapes, uint32_t axis )
{
    uint4* out = reinterpret_cast< uint4* >( nodes + 8ull * slot );
    out[0]     = uint4{ __float_as_uint( box.lo[0] ), __float_as_uint( box.lo[1] ), __float_as_uint( box.lo[2] ), __float_as_uint( box.hi[0] ) };
    out[1]     = uint4{ __float_as_uint( box.hi[1] ), __float_as_uint( box.hi[2] ), offset, numShapes | ( axis << 16 ) };
}

// Position of this thread's run among the runs of equal `r` of the block's consecutive threads, the number of runs
// in the block, and whether the thread starts its run.  All threads of the block call it.
__device__ __forceinline__ uint32_t RunIndex( uint32_t r, uint32_t* sLast, uint32_t* sHeads, uint32_t* runs, bool* head )
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t before = __shfl_up( r, 1 );
    if ( lane == 63 ) sLast[wave] = r;
    __syncthreads();
    if ( lane == 0 ) before = wave ? sLast[wave - 1] : ~r;
    const bool isHead   = r != before;
    const uint64_t mask = __ballot( isHead );
    const uint32_t upTo = static_cast< uint32_t >( __popcll( mask & ( ( 2ull << lane ) - 1ull ) ) );
    if ( lane == 0 ) sHeads[wave] = static_cast< uint32_t >( __popcll( mask ) );
    __syncthreads();
    uint32_t base = 0, total = 0;
    for ( uint32_t w = 0; w < kWaves; ++w )
    {
        const uint32_t c = sHeads[w];
        if ( w < wave ) base += c;
        total += c;
    }
    *runs = total;
    *head = isHead;
    return base + upTo - 1u;
}

// Only for ranges whose box the level before could not hand down (the root; the halves of a range that was halved
// where it lay): every shape of such a range adds its box and centroid to the range's sums.
__global__ __launch_bounds__( kBlock ) void ReduceKernel( const float4* __restrict__ items, const uint32_t* __restrict__ seg, BigRange* __restrict__ cur, uint32_t n )
{
    __shared__ uint32_t sLast[kWaves], sHeads[kWaves];
    __shared__ uint32_t sAcc[kMaxRuns][12];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    uint32_t r       = i < n ? seg[i] : kNone;
    if ( r != kNone && !cur[r].needReduce ) r = kNone;
    uint32_t v[12];
    if ( r != kNone )
    {
        const float4 a = items[2ull * i], b = items[2ull * i + 1];
        v[0] = Enc( a.x ); v[1] = Enc( a.y ); v[2] = Enc( a.z );
        v[6] = Enc( b.x ); v[7] = Enc( b.y ); v[8] = Enc( b.z );
        v[3] = v[9]  = Enc( Centroid( a.x, b.x ) );
        v[4] = v[10] = Enc( Centroid( a.y, b.y ) );
        v[5] = v[11] = Enc( Centroid( a.z, b.z ) );
    }
    uint32_t runs;
    bool head;
    const uint32_t run = RunIndex( r, sLast, sHeads, &runs, &head );
    if ( runs <= kMaxRuns )
    {
        if ( head )
            for ( int k = 0; k < 12; ++k ) sAcc[run][k] = k < 6 ? 0xFFFFFFFFu : 0u;
        __syncthreads();
        if ( r != kNone )
        {
            for ( int k = 0; k < 6; ++k ) atomicMin( &sAcc[run][k], v[k] );
            for ( int k = 6; k < 12; ++k ) atomicMax( &sAcc[run][k], v[k] );
        }
        __syncthreads();
        if ( head && r != kNone )
        {
            for ( int k = 0; k < 6; ++k ) atomicMin( &cur[r].acc[k], sAcc[run][k] );
            for ( int k = 6; k < 12; ++k ) atomicMax( &cur[r].acc[k], sAcc[run][k] );
        }
    }
    else if ( r != kNone )
    {
        for ( int k = 0; k < 6; ++k ) atomicMin( &cur[r].acc[k], v[k] );
        for ( int k = 6; k < 12; ++k ) atomicMax( &cur[r].acc[k], v[k] );
    }
}

// What a range needs before its shapes can be sorted into buckets: split axis (src/bvh.cpp:68-70), the centroid
// bounds along it, the area of its box - and its node record, which gets the second child's slot in SplitKernel.
__device__ __forceinline__ void PrepareRange( BigRange& R, const Box& box, const Box& cbox, uint32_t* nodes )
{
    const int dim = LongestDimension( cbox );
    R.dim         = static_cast< uint32_t >( dim );
    R.cMin        = cbox.lo[dim];
    R.cMax        = cbox.hi[dim];
    R.area        = SurfaceArea( box );
    WriteNode( nodes, R.slot, box, 0u, 0u, static_cast< uint32_t >( dim ) );
}

// The ranges ReduceKernel summed: box and centroid box from the sums.
__global__ void AxisKernel( BigRange* __restrict__ cur, uint32_t count, uint32_t* __restrict__ nodes )
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if ( r >= count || !cur[r].needReduce ) return;
    BigRange& R = cur[r];
    Box box, cbox;
    for ( int k = 0; k < 3; ++k )
    {
        box.lo[k]  = Dec( R.acc[k] );
        cbox.lo[k] = Dec( R.acc[3 + k] );
        box.hi[k]  = Dec( R.acc[6 + k] );
        cbox.hi[k] = Dec( R.acc[9 + k] );
    }
    PrepareRange( R, box, cbox, nodes );
}

// Every shape finds the range it belongs to now: its old range's first or second half.
__global__ __launch_bounds__( kBlock ) void SegUpdateKernel( uint32_t* __restrict__ seg, const BigRange* __restrict__ prev, uint32_t n )
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if ( i >= n ) return;
    const uint32_t r = seg[i];
    if ( r == kNone ) return;
    const BigRange* p = prev + r;
    seg[i]            = i < p->cutoff ? p->left : p->right;
}

__global__ void InitBucketsKernel( uint32_t* __restrict__ words, uint32_t count )
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if ( k >= count ) return;
    const uint32_t word = k % kBucketWords;
    words[k]            = ( word >= 1 && word <= 6 ) ? 0xFFFFFFFFu : 0u; // count 0, minima at the top, maxima at the bottom
}

// One shape into the sums of its bucket in LDS.  A bound that does not lower a minimum (raise a maximum) as it reads
// now cannot do so later either, the sums only move one way: the plain read first spares most of the atomics, which
// pile up on few addresses otherwise.
__device__ __forceinline__ void AddToBucket( uint32_t* w, const uint32_t* v )
{
    atomicAdd( w, 1u );
    const volatile uint32_t* now = w;
    for ( int k = 0; k < 6; ++k )
        if ( v[k] < now[1 + k] ) atomicMin( w + 1 + k, v[k] );
    for ( int k = 6; k < 12; ++k )
        if ( v[k] > now[1 + k] ) atomicMax( w + 1 + k, v[k] );
}

// Level step 1 (src/bvh.cpp:104-112, and for the level after: src/bvh.cpp:41-45,62-70): every shape moves on to its
// half of its old range (updateSeg), finds its bucket along its range's split axis and adds itself to the bucket's
// count, box and centroid box.  The unions of the buckets on either side of the split that SplitKernel picks are
// the boxes and centroid boxes of the two halves: the next level starts with them and sums nothing else.
//
// A block takes `tiles` consecutive tiles of kBlock shapes.  Ranges are intervals of positions, so if the first and
// the last shape of the block's tiles are in the same range, all of them are: then every wave sums into its own
// copy of the twelve buckets in LDS and the block adds the copies to the range's sums once.  Otherwise each tile is
// cut into its runs of equal range, at most kMaxRuns of them summed in LDS, more than that straight in memory.
__global__ __launch_bounds__( kBlock ) void BucketKernel( const float4* __restrict__ items, uint32_t* __restrict__ seg, const BigRange* __restrict__ prev,
                                                           const BigRange* __restrict__ cur, Buckets* __restrict__ buckets, uint32_t n, uint32_t tiles,
                                                           int updateSeg )
{
    __shared__ uint32_t sLast[kWaves], sHeads[kWaves], sRunRange[kMaxRuns], sEnds[2];
    __shared__ uint32_t sBk[kMaxRuns * kBuckets * kBucketWords]; // [run or wave][bucket][word]
    const uint32_t first = blockIdx.x * tiles * kBlock;
    const uint32_t wave  = threadIdx.x >> 6;
    auto rangeOf = [&]( uint32_t i ) -> uint32_t {
        uint32_t r = seg[i];
        if ( updateSeg && r != kNone )
        {
            const BigRange* p = prev + r;
            r                 = i < p->cutoff ? p->left : p->right;
        }
        return r;
    };
    if ( threadIdx.x < 2 )
    {
        const uint32_t lastInBlock = first + tiles * kBlock - 1u;
        sEnds[threadIdx.x]         = rangeOf( threadIdx.x == 0 ? first : ( lastInBlock < n ? lastInBlock : n - 1u ) );
    }
    __syncthreads();
    const uint32_t only = sEnds[0];
    const bool oneRange = only == sEnds[1] && only != kNone;
    if ( oneRange )
    {
        for ( uint32_t k = threadIdx.x; k < kWaves * kBuckets * kBucketWords; k += kBlock )
        {
            const uint32_t word = k % kBucketWords;
            sBk[k]              = ( word >= 1 && word <= 6 ) ? 0xFFFFFFFFu : 0u;
        }
        __syncthreads();
        const BigRange* R  = cur + only;
        const uint32_t dim = R->dim & 0xFFu;
        const float cMin = R->cMin, cMax = R->cMax;
        for ( uint32_t t = 0; t < tiles; ++t )
        {
            const uint32_t i = first + t * kBlock + threadIdx.x;
            if ( i >= n ) break;
            if ( updateSeg ) seg[i] = only;
            const float4 lo = items[2ull * i], hi = items[2ull * i + 1];
            const float cx = Centroid( lo.x, hi.x ), cy = Centroid( lo.y, hi.y ), cz = Centroid( lo.z, hi.z );
            const int b    = BucketOf( dim == 0 ? cx : dim == 1 ? cy : cz, cMin, cMax );
            uint32_t* w    = sBk + ( wave * kBuckets + b ) * kBucketWords;
            const uint32_t v[12] = { Enc( lo.x ), Enc( lo.y ), Enc( lo.z ), Enc( cx ), Enc( cy ), Enc( cz ),
                                     Enc( hi.x ), Enc( hi.y ), Enc( hi.z ), Enc( cx ), Enc( cy ), Enc( cz ) };
            AddToBucket( w, v );
        }
        __syncthreads();
        if ( threadIdx.x < kBuckets * kBucketWords )
        {
            const uint32_t b = threadIdx.x / kBucketWords, word = threadIdx.x % kBucketWords;
            uint32_t total = 0;
            for ( uint32_t w = 0; w < kWaves; ++w ) total += sBk[( w * kBuckets + b ) * kBucketWords];
            if ( total && word < 13 )
            {
                uint32_t value = word == 0 ? 0u : word <= 6 ? 0xFFFFFFFFu : 0u;
                for ( uint32_t w = 0; w < kWaves; ++w )
                {
                    const uint32_t x = sBk[( w * kBuckets + b ) * kBucketWords + word];
                    value            = word == 0 ? value + x : word <= 6 ? ( x < value ? x : value ) : ( x > value ? x : value );
                }
                uint32_t* g = buckets[only].w[b] + word;
                if ( word == 0 ) atomicAdd( g, value );
                else if ( word <= 6 ) atomicMin( g, value );
                else atomicMax( g, value );
            }
        }
        return;
    }
    for ( uint32_t t = 0; t < tiles; ++t )
    {
        const uint32_t i = first + t * kBlock + threadIdx.x;
        if ( first + t * kBlock >= n ) break; // the whole tile lies behind the last shape
        const uint32_t r = i < n ? rangeOf( i ) : kNone;
        if ( updateSeg && i < n ) seg[i] = r;
        uint32_t v[12];
        int b = 0;
        if ( r != kNone )
        {
            const float4 lo = items[2ull * i], hi = items[2ull * i + 1];
            const BigRange* R = cur + r;
            const uint32_t dim = R->dim & 0xFFu;
            const float cx = Centroid( lo.x, hi.x ), cy = Centroid( lo.y, hi.y ), cz = Centroid( lo.z, hi.z );
            b    = BucketOf( dim == 0 ? cx : dim == 1 ? cy : cz, R->cMin, R->cMax );
            v[0] = Enc( lo.x ); v[1] = Enc( lo.y ); v[2] = Enc( lo.z ); v[3] = Enc( cx ); v[4] = Enc( cy ); v[5] = Enc( cz );
            v[6] = Enc( hi.x ); v[7] = Enc( hi.y ); v[8] = Enc( hi.z ); v[9] = v[3]; v[10] = v[4]; v[11] = v[5];
        }
        uint32_t runs;
        bool head;
        const uint32_t run = RunIndex( r, sLast, sHeads, &runs, &head );
        if ( runs <= kMaxRuns )
        {
            for ( uint32_t k = threadIdx.x; k < runs * kBuckets * kBucketWords; k += kBlock )
            {
                const uint32_t word = k % kBucketWords;
                sBk[k]              = ( word >= 1 && word <= 6 ) ? 0xFFFFFFFFu : 0u;
            }
            if ( head ) sRunRange[run] = r;
            __syncthreads();
            if ( r != kNone )
            {
                uint32_t* w = sBk + ( run * kBuckets + b ) * kBucketWords;
                AddToBucket( w, v );
            }
            __syncthreads();
            for ( uint32_t k = threadIdx.x; k < runs * kBuckets * kBucketWords; k += kBlock )
            {
                const uint32_t word = k % kBucketWords, q = k / kBucketWords % kBuckets, owner = sRunRange[k / ( kBuckets * kBucketWords )];
                if ( owner == kNone || word >= 13 || sBk[k - word] == 0 ) continue;
                uint32_t* g = buckets[owner].w[q] + word;
                if ( word == 0 ) atomicAdd( g, sBk[k] );
                else if ( word <= 6 ) atomicMin( g, sBk[k] );
                else atomicMax( g, sBk[k] );
            }
            __syncthreads(); // the next tile clears what this one still reads
        }
        else if ( r != kNone )
        {
            uint32_t* g = buckets[r].w[b];
            atomicAdd( g, 1u );
            for ( int k = 0; k < 6; ++k ) atomicMin( g + 1 + k, v[k] );
            for ( int k = 6; k < 12; ++k ) atomicMax( g + 1 + k, v[k] );
        }
    }
}

// The cheapest of the eleven splits between the twelve buckets (src/bvh.cpp:114-147).  The union of buckets 0..i is
// built up as i grows, which is the reference's own order of operations; the union of buckets i+1..11 from the far
// end, which gives the same bounds.
__device__ __forceinline__ int CheapestSplit( const Box* bb, const int* cnt, float area )
{
    float areaBehind[kBuckets - 1];
    int countBehind[kBuckets - 1];
    Box b1 = EmptyBox();
    int count1 = 0;
    for ( int i = kBuckets - 2; i >= 0; --i )
    {
        Union( b1, bb[i + 1] );
        count1 += cnt[i + 1];
        areaBehind[i]  = SurfaceArea( b1 );
        countBehind[i] = count1;
    }
    float minCost = 0;
    int minBucket = 0, count0 = 0;
    Box b0 = EmptyBox();
    for ( int i = 0; i < kBuckets - 1; ++i )
    {
        Union( b0, bb[i] );
        count0 += cnt[i];
        const float cost = 0.5f + ( count0 * SurfaceArea( b0 ) + countBehind[i] * areaBehind[i] ) / area;
        if ( i == 0 || cost < minCost )
        {
            minCost   = cost;
            minBucket = i;
        }
    }
    return minBucket;
}

// A half of a split range becomes a range of the next level (with its box and centroid box, if the split knows
// them) or, if it is small, the work of one thread.
__device__ __forceinline__ void NewRange( uint32_t start, uint32_t end, uint32_t slot, bool known, const Box& box, const Box& cbox, uint32_t* nodes,
                                          BigRange* next, uint32_t* counters, SmallRange* small, uint32_t* id )
{
    if ( end - start > kThreadRange )
    {
        const uint32_t k = atomicAdd( counters + 0, 1u );
        BigRange& R      = next[k];
        R.start = start; R.end = end; R.slot = slot; R.cutoff = start;
        R.left = R.right = kNone;
        R.dim = 0; R.noMove = 0;
        R.needReduce = known ? 0u : 1u;
        for ( int q = 0; q < 12; ++q ) R.acc[q] = q < 6 ? 0xFFFFFFFFu : 0u;
        if ( known ) PrepareRange( R, box, cbox, nodes );
        else atomicAdd( counters + 2, 1u );
        *id = k;
    }
    else
    {
        const uint32_t k = atomicAdd( counters + 1, 1u );
        small[k]         = SmallRange{ start, end, slot };
        *id              = kNone;
    }
}

// Level step 2: the eleven split costs, the cheapest split, the partition point and the two ranges of the next
// level (src/bvh.cpp:114-169).
__global__ void SplitKernel( BigRange* __restrict__ cur, const Buckets* __restrict__ buckets, uint32_t count, uint32_t* __restrict__ nodes, BigRange* __restrict__ next,
                             SmallRange* __restrict__ small, uint32_t* __restrict__ counters )
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if ( r >= count ) return;
    BigRange& R = cur[r];
    Box bb[kBuckets], cb[kBuckets];
    int cnt[kBuckets];
    for ( int b = 0; b < kBuckets; ++b )
    {
        const uint32_t* w = buckets[r].w[b];
        cnt[b]            = static_cast< int >( w[0] );
        // an empty bucket is the empty box: Dec( 0xFFFFFFFF ) is a NaN, so spell it out
        for ( int k = 0; k < 3; ++k )
        {
            bb[b].lo[k] = cnt[b] ? Dec( w[1 + k] ) : kFltMax;
            cb[b].lo[k] = cnt[b] ? Dec( w[4 + k] ) : kFltMax;
            bb[b].hi[k] = cnt[b] ? Dec( w[7 + k] ) : -kFltMax;
            cb[b].hi[k] = cnt[b] ? Dec( w[10 + k] ) : -kFltMax;
        }
    }
    const int minBucket = CheapestSplit( bb, cnt, R.area );
    uint32_t passing = 0;
    for ( int j = 0; j <= minBucket; ++j ) passing += static_cast< uint32_t >( cnt[j] );
    uint32_t cutoff = R.start + passing;
    R.dim |= static_cast< uint32_t >( minBucket ) << 8;
    bool known = true;
    if ( cutoff == R.start || cutoff == R.end )
    {
        // every shape on one side: std::partition moved nothing; the range is halved where it lies (the reference
        // itself recurses without end here, src/bvh.cpp:173-175) and the halves' boxes have to be summed
        cutoff   = static_cast< uint32_t >( ( static_cast< uint64_t >( R.start ) + R.end ) / 2u );
        R.noMove = 1;
        known    = false;
    }
    Box box0 = EmptyBox(), cbox0 = EmptyBox(), box1 = EmptyBox(), cbox1 = EmptyBox();
    for ( int j = 0; j < kBuckets; ++j )
    {
        Union( j <= minBucket ? box0 : box1, bb[j] );
        Union( j <= minBucket ? cbox0 : cbox1, cb[j] );
    }
    R.cutoff              = cutoff;
    const uint32_t second = R.slot + 2u * ( cutoff - R.start );
    nodes[8ull * R.slot + 6] = second;
    NewRange( R.start, cutoff, R.slot + 1u, known, box0, cbox0, nodes, next, counters, small, &R.left );
    NewRange( cutoff, R.end, second, known, box1, cbox1, nodes, next, counters, small, &R.right );
}

// Level step 5: the predicate of std::partition (src/bvh.cpp:164-168) per shape, as the input of a prefix sum.
__global__ __launch_bounds__( kBlock ) void FlagKernel( const float4* __restrict__ items, const uint32_t* __restrict__ seg, const BigRange* __restrict__ cur,
                                                         uint32_t* __restrict__ prefix, uint32_t n, uint32_t padded )
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if ( i >= padded ) return;
    uint32_t flag = 0;
    if ( i < n )
    {
        const uint32_t r = seg[i];
        if ( r != kNone )
        {
            const BigRange* R = cur + r;
            if ( !R->noMove )
            {
                const uint32_t dim = R->dim & 0xFFu, minBucket = R->dim >> 8;
                const float4 lo = items[2ull * i], hi = items[2ull * i + 1];
                const float c = dim == 0 ? Centroid( lo.x, hi.x ) : dim == 1 ? Centroid( lo.y, hi.y ) : Centroid( lo.z, hi.z );
                flag = static_cast< uint32_t >( BucketOf( c, R->cMin, R->cMax ) ) <= minBucket ? 1u : 0u;
            }
        }
    }
    prefix[i] = flag;
}

// --- exclusive prefix sum over `padded` (a multiple of 4 * kBlock) words, in place --------------------------------

__device__ __forceinline__ uint32_t BlockExclusive( uint32_t value, uint32_t* sWave, uint32_t* total )
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = value;
    for ( int d = 1; d < 64; d <<= 1 )
    {
        const uint32_t up = __shfl_up( incl, d );
        if ( lane >= static_cast< uint32_t >( d ) ) incl += up;
    }
    if ( lane == 63 ) sWave[wave] = incl;
    __syncthreads();
    uint32_t base = 0, sum = 0;
    for ( uint32_t w = 0; w < kWaves; ++w )
    {
        const uint32_t c = sWave[w];
        if ( w < wave ) base += c;
        sum += c;
    }
    __syncthreads();
    *total = sum;
    return base + incl - value;
}

__global__ __launch_bounds__( kBlock ) void ScanSumsKernel( const uint4* __restrict__ data, uint32_t* __restrict__ blockSums )
{
    __shared__ uint32_t sWave[kWaves];
    const uint4 q = data[static_cast< uint64_t >( blockIdx.x ) * kBlock + threadIdx.x];
    uint32_t total;
    (void)BlockExclusive( q.x + q.y + q.z + q.w, sWave, &total );
    if ( threadIdx.x == 0 ) blockSums[blockIdx.x] = total;
}

__global__ __launch_bounds__( kBlock ) void ScanTopKernel( uint32_t* __restrict__ blockSums, uint32_t count )
{
    __shared__ uint32_t sWave[kWaves];
    const uint32_t per = ( count + kBlock - 1 ) / kBlock;
    const uint32_t begin = threadIdx.x * per, end = begin + per < count ? begin + per : count;
    uint32_t sum = 0;
    for ( uint32_t k = begin; k < end; ++k ) sum += blockSums[k];
    uint32_t total;
    uint32_t running = BlockExclusive( sum, sWave, &total );
    for ( uint32_t k = begin; k < end; ++k )
    {
        const uint32_t x = blockSums[k];
        blockSums[k]     = running;
        running += x;
    }
}

__global__ __launch_bounds__( kBlock ) void ScanApplyKernel( uint4* __restrict__ data, const uint32_t* __restrict__ blockSums )
{
    __shared__ uint32_t sWave[kWaves];
    const uint64_t at = static_cast< uint64_t >( blockIdx.x ) * kBlock + threadIdx.x;
    const uint4 q     = data[at];
    uint32_t total;
    const uint32_t base = blockSums[blockIdx.x] + BlockExclusive( q.x + q.y + q.z + q.w, sWave, &total );
    data[at]            = uint4{ base, base + q.x, base + q.x + q.y, base + q.x + q.y + q.z };
}

// Level step 6: who changes places with whom.  Within a range [s, e) with partition point c, the k-th shape before c
// that fails the predicate and the k-th shape from the right behind c that passes it are a pair.
__global__ __launch_bounds__( kBlock ) void PartnerKernel( const uint32_t* __restrict__ seg, const BigRange* __restrict__ cur, const uint32_t* __restrict__ prefix,
                                                            uint32_t* __restrict__ failing, uint32_t* __restrict__ passing, uint32_t n )
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if ( i >= n ) return;
    const uint32_t r = seg[i];
    if ( r == kNone ) return;
    const BigRange* R = cur + r;
    if ( R->noMove ) return;
    const uint32_t s = R->start, c = R->cutoff;
    const uint32_t before = prefix[i] - prefix[s];
    const bool pass       = prefix[i + 1] != prefix[i];
    if ( i < c && !pass ) failing[s + ( i - s ) - before] = i;
    if ( i >= c && pass ) passing[s + ( c - s ) - before - 1u] = i;
}

// Level step 7: the pairs change places.
__global__ __launch_bounds__( kBlock ) void SwapKernel( float4* __restrict__ items, const uint32_t* __restrict__ seg, const BigRange* __restrict__ cur,
                                                         const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ failing, const uint32_t* __restrict__ passing,
                                                         uint32_t n )
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if ( i >= n ) return;
    const uint32_t r = seg[i];
    if ( r == kNone ) return;
    const BigRange* R = cur + r;
    if ( R->noMove ) return;
    const uint32_t s = R->start, c = R->cutoff;
    const uint32_t pairs = ( c - s ) - ( prefix[c] - prefix[s] );
    if ( i - s >= pairs ) return;
    const uint64_t a = failing[i], b = passing[i];
    const float4 a0 = items[2 * a], a1 = items[2 * a + 1], b0 = items[2 * b], b1 = items[2 * b + 1];
    items[2 * a] = b0; items[2 * a + 1] = b1;
    items[2 * b] = a0; items[2 * b + 1] = a1;
}

// --- small ranges: one thread finishes the subtree -------------------------------------------------

struct Shape
{
    Box box;
    uint32_t index;
};

__device__ __forceinline__ float Key( const Shape& s, int dim ) { return Centroid( s.box.lo[dim], s.box.hi[dim] ); }

// libstdc++ (GCC 11) bits/stl_algo.h: __insertion_sort, __move_median_to_first, __unguarded_partition,
// __introselect as std::nth_element runs them on at most four elements (the depth limit is never reached).
__device__ void InsertionSort( Shape* v, int first, int last, int dim )
{
    for ( int i = first + 1; i < last; ++i )
    {
        const Shape val = v[i];
        const float key = Key( val, dim );
        int j           = i;
        while ( j > first && key < Key( v[j - 1], dim ) )
        {
            v[j] = v[j - 1];
            --j;
        }
        v[j] = val;
    }
}

__device__ __forceinline__ void SwapShapes( Shape* v, int a, int b )
{
    const Shape t = v[a];
    v[a]          = v[b];
    v[b]          = t;
}

__device__ void NthElement( Shape* v, int first, int nth, int last, int dim )
{
    while ( last - first > 3 )
    {
        const int mid = first + ( last - first ) / 2;
        const int a = first + 1, b = mid, c = last - 1;
        const float ka = Key( v[a], dim ), kb = Key( v[b], dim ), kc = Key( v[c], dim );
        if ( ka < kb )
        {
            if ( kb < kc ) SwapShapes( v, first, b );
            else if ( ka < kc ) SwapShapes( v, first, c );
            else SwapShapes( v, first, a );
        }
        else if ( ka < kc ) SwapShapes( v, first, a );
        else if ( kb < kc ) SwapShapes( v, first, c );
        else SwapShapes( v, first, b );

        const float pivot = Key( v[first], dim );
        int lo = first + 1, hi = last;
        while ( true )
        {
            while ( lo < last && Key( v[lo], dim ) < pivot ) ++lo;
            --hi;
            while ( hi > first && pivot < Key( v[hi], dim ) ) --hi;
            if ( !( lo < hi ) ) break;
            SwapShapes( v, lo, hi );
            ++lo;
        }
        if ( lo <= nth ) first = lo;
        else last = lo;
    }
    InsertionSort( v, first, last, dim );
}

// One thread builds the subtree over a range of at most kThreadRange shapes the way the reference's recursion does:
// boxes in shape order, the bucketed split with std::partition's own two-pointer walk (libstdc++, bits/stl_algo.h
// __partition for bidirectional iterators), nth_element on ranges of at most four.  Of the two halves of a range
// the smaller is built first while the larger waits, so that few ranges ever wait at once.  The thread's shapes
// are copied to LDS first (word w of shape k of lane l at ( k * 7 + w ) * 64 + l: no two lanes share a bank more
// than any 64-lane access must) and sorted there - a thread walks its range many times, and 64 threads doing that
// 1 KB apart in memory would fetch a cache line for every 16 bytes they use.  Nothing is written back but nodes and
// shape order: nobody reads the shapes of a finished range again.
constexpr int kSubtreeWords = 7; // lo xyz, hi xyz, shape index
__global__ __launch_bounds__( 64 ) void SubtreeKernel( const float4* __restrict__ items, const SmallRange* __restrict__ small, uint32_t count,
                                                        uint32_t* __restrict__ nodes, uint32_t* __restrict__ order )
{
    __shared__ uint32_t sShapes[kThreadRange * kSubtreeWords * 64];
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if ( r >= count ) return;
    uint32_t* mine = sShapes + threadIdx.x;
    auto word  = [&]( uint32_t k, int w ) -> uint32_t& { return mine[( k * kSubtreeWords + w ) * 64u]; };
    auto boxOf = [&]( uint32_t k ) {
        return Box{ { __uint_as_float( word( k, 0 ) ), __uint_as_float( word( k, 1 ) ), __uint_as_float( word( k, 2 ) ) },
                    { __uint_as_float( word( k, 3 ) ), __uint_as_float( word( k, 4 ) ), __uint_as_float( word( k, 5 ) ) } };
    };
    const uint32_t base = small[r].start, total = small[r].end - small[r].start;
    for ( uint32_t k = 0; k < total; ++k )
    {
        const float4 lo = items[2ull * ( base + k )], hi = items[2ull * ( base + k ) + 1];
        word( k, 0 ) = __float_as_uint( lo.x ); word( k, 1 ) = __float_as_uint( lo.y ); word( k, 2 ) = __float_as_uint( lo.z );
        word( k, 3 ) = __float_as_uint( hi.x ); word( k, 4 ) = __float_as_uint( hi.y ); word( k, 5 ) = __float_as_uint( hi.z );
        word( k, 6 ) = __float_as_uint( lo.w );
    }
    constexpr int kWaiting = 8; // log2( kThreadRange ) + 1 is enough
    uint32_t waitS[kWaiting], waitE[kWaiting], waitSlot[kWaiting];
    int waiting = 0;
    uint32_t s = 0, e = total, slot = small[r].slot; // positions relative to the range's first shape
    while ( true )
    {
        Box box = EmptyBox(), cbox = EmptyBox();
        for ( uint32_t i = s; i < e; ++i )
        {
            const Box b = boxOf( i );
            Union( box, b );
            Box c;
            for ( int q = 0; q < 3; ++q ) c.lo[q] = c.hi[q] = Centroid( b.lo[q], b.hi[q] );
            Union( cbox, c );
        }
        if ( e - s == 1 ) // src/bvh.cpp:47-56
        {
            WriteNode( nodes, slot, box, base + s, 1u, 0u );
            order[base + s] = word( s, 6 );
            if ( waiting == 0 ) break;
            --waiting;
            s = waitS[waiting]; e = waitE[waiting]; slot = waitSlot[waiting];
            continue;
        }
        const int dim = LongestDimension( cbox );
        uint32_t cutoff;
        if ( e - s <= kSmallRange ) // src/bvh.cpp:95-98: the middle element by nth_element
        {
            const int n = static_cast< int >( e - s );
            Shape v[kSmallRange];
            for ( int k = 0; k < n; ++k )
            {
                v[k].box   = boxOf( s + k );
                v[k].index = word( s + k, 6 );
            }
            // the middle of the range counted from the array's start: ( start + end ) / 2 with start = base + s, end = base + e
            cutoff = static_cast< uint32_t >( ( 2ull * base + s + e ) / 2u - base );
            NthElement( v, 0, static_cast< int >( cutoff - s ), n, dim );
            for ( int k = 0; k < n; ++k )
            {
                for ( int q = 0; q < 3; ++q )
                {
                    word( s + k, q )     = __float_as_uint( v[k].box.lo[q] );
                    word( s + k, 3 + q ) = __float_as_uint( v[k].box.hi[q] );
                }
                word( s + k, 6 ) = v[k].index;
            }
        }
        else // src/bvh.cpp:100-169
        {
            Box bb[kBuckets];
            int cnt[kBuckets];
            for ( int b = 0; b < kBuckets; ++b )
            {
                bb[b]  = EmptyBox();
                cnt[b] = 0;
            }
            const float cMin = cbox.lo[dim], cMax = cbox.hi[dim];
            auto bucketOf = [&]( uint32_t i ) {
                return BucketOf( Centroid( __uint_as_float( word( i, dim ) ), __uint_as_float( word( i, 3 + dim ) ) ), cMin, cMax );
            };
            for ( uint32_t i = s; i < e; ++i )
            {
                const int q = bucketOf( i );
                cnt[q]++;
                Union( bb[q], boxOf( i ) );
            }
            const int minBucket = CheapestSplit( bb, cnt, SurfaceArea( box ) );
            uint32_t first = s, last = e;
            while ( true )
            {
                while ( first != last && bucketOf( first ) <= minBucket ) ++first;
                if ( first == last ) break;
                --last;
                while ( first != last && !( bucketOf( last ) <= minBucket ) ) --last;
                if ( first == last ) break;
                for ( int w = 0; w < kSubtreeWords; ++w )
                {
                    const uint32_t t = word( first, w );
                    word( first, w ) = word( last, w );
                    word( last, w )  = t;
                }
                ++first;
            }
            cutoff = first;
            if ( cutoff == s || cutoff == e ) cutoff = static_cast< uint32_t >( ( 2ull * base + s + e ) / 2u - base ); // as in SplitKernel
        }
        const uint32_t second = slot + 2u * ( cutoff - s );
        WriteNode( nodes, slot, box, second, 0u, static_cast< uint32_t >( dim ) );
        if ( cutoff - s <= e - cutoff )
        {
            waitS[waiting] = cutoff; waitE[waiting] = e; waitSlot[waiting] = second;
            ++waiting;
            e    = cutoff;
            slot = slot + 1u;
        }
        else
        {
            waitS[waiting] = s; waitE[waiting] = cutoff; waitSlot[waiting] = slot + 1u;
            ++waiting;
            s    = cutoff;
            slot = second;
        }
    }
}

// Shapes in scene order; *notFinite is set if a bound is an infinity or a NaN (the sums above order floats as
// integers, which has no place for a NaN), or if a centre - half the sum of two bounds, as the reference computes it
// (src/bvh.cpp:213) - would overflow to one.
__global__ void FillItemsKernel( const float* __restrict__ boxes, float4* __restrict__ items, uint32_t n, uint32_t* __restrict__ notFinite )
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    const float* b   = boxes + 6ull * i;
    items[2ull * i]     = float4{ b[0], b[1], b[2], __uint_as_float( i ) };
    items[2ull * i + 1] = float4{ b[3], b[4], b[5], 0.0f };
    bool finite = true;
    for ( int k = 0; k < 6; ++k ) finite = finite && fabsf( b[k] ) <= kFltMax;
    for ( int k = 0; k < 3; ++k ) finite = finite && fabsf( b[k] + b[k + 3] ) <= kFltMax;
    if ( !finite ) atomicOr( notFinite, 1u );
}

struct Dev
{
    void* p = nullptr;
    ~Dev()
    {
        if ( p ) (void)hipFree( p );
    }
    hipError_t Alloc( size_t bytes ) { return hipMalloc( &p, bytes ? bytes : 4 ); }
    template < typename T >
    T* As() const { return static_cast< T* >( p ); }
};

int Fail( int code, const char* what, hipError_t e = hipSuccess )
{
    char buf[256];
    if ( e != hipSuccess )
    {
        (void)hipGetLastError();
        snprintf( buf, sizeof( buf ), "pt_build_bvh: %s: %s", what, hipGetErrorString( e ) );
    }
    else
        snprintf( buf, sizeof( buf ), "pt_build_bvh: %s", what );
    return ptd::SetError( code, buf );
}

#define BVH_TRY( expr )                                                     \
    do                                                                      \
    {                                                                       \
        const hipError_t e_ = ( expr );                                     \
        if ( e_ != hipSuccess ) return Fail( PT_ERR_DEVICE, #expr, e_ ); \
    } while ( 0 )

int Build( int32_t device, const float* boxes, uint64_t n64, PtBvhNode* nodesOut, uint32_t* orderOut, uint64_t* numNodesOut, float* msOut )
{
    if ( !boxes || !nodesOut || !orderOut || n64 == 0 ) return Fail( PT_ERR_INVALID_ARG, "null pointer or no shapes" );
    if ( n64 > 0x3FFFFFFFull ) return Fail( PT_ERR_INVALID_ARG, "too many shapes for 32-bit node offsets" );
    const uint32_t n      = static_cast< uint32_t >( n64 );
    const uint32_t padded = ( ( n + 1u + 4u * kBlock - 1u ) / ( 4u * kBlock ) ) * ( 4u * kBlock ); // prefix sums: n + 1 words, whole blocks of uint4
    const uint32_t scanBlocks = padded / ( 4u * kBlock );
    const uint32_t bigMax     = n / ( kThreadRange + 1u ) + 1u;
    const uint32_t itemBlocks = ( n + kBlock - 1u ) / kBlock;
    // tiles of kBlock shapes per block of BucketKernel: many, so that few blocks add to the sums of the big ranges near
    // the root, but not so many that the GPU's 256 compute units run out of blocks
    const uint32_t tiles        = n >= ( 16u << 20 ) ? 16u : n >= ( 1u << 20 ) ? n >> 20 : 1u;
    const uint32_t bucketBlocks = ( n + tiles * kBlock - 1u ) / ( tiles * kBlock );

    int deviceCount = 0;
    BVH_TRY( hipGetDeviceCount( &deviceCount ) );
    if ( device < 0 || device >= deviceCount ) return Fail( PT_ERR_INVALID_ARG, "no such device" );
    BVH_TRY( hipSetDevice( device ) );

    Dev dBoxes, dItems, dSeg, dPrefix, dSums, dFailing, dPassing, dRanges[2], dBuckets, dSmall, dCounters, dNodes, dOrder;
    BVH_TRY( dBoxes.Alloc( 24ull * n ) );
    BVH_TRY( dItems.Alloc( 32ull * n ) );
    BVH_TRY( dSeg.Alloc( 4ull * n ) );
    BVH_TRY( dPrefix.Alloc( 4ull * padded ) );
    BVH_TRY( dSums.Alloc( 4ull * scanBlocks ) );
    BVH_TRY( dFailing.Alloc( 4ull * n ) );
    BVH_TRY( dPassing.Alloc( 4ull * n ) );
    BVH_TRY( dRanges[0].Alloc( sizeof( BigRange ) * static_cast< size_t >( bigMax ) ) );
    BVH_TRY( dRanges[1].Alloc( sizeof( BigRange ) * static_cast< size_t >( bigMax ) ) );
    BVH_TRY( dBuckets.Alloc( sizeof( Buckets ) * static_cast< size_t >( bigMax ) ) );
    BVH_TRY( dSmall.Alloc( sizeof( SmallRange ) * static_cast< size_t >( n ) ) );
    BVH_TRY( dCounters.Alloc( 12 ) );
    BVH_TRY( dNodes.Alloc( 32ull * ( 2ull * n - 1ull ) ) );
    BVH_TRY( dOrder.Alloc( 4ull * n ) );

    hipStream_t stream = nullptr; // the default stream: every step depends on the one before
    hipEvent_t t0 = nullptr, t1 = nullptr;
    BVH_TRY( hipEventCreate( &t0 ) );
    BVH_TRY( hipEventCreate( &t1 ) );
    struct Events
    {
        hipEvent_t a, b;
        ~Events()
        {
            (void)hipEventDestroy( a );
            (void)hipEventDestroy( b );
        }
    } events{ t0, t1 };

    BVH_TRY( hipMemcpy( dBoxes.p, boxes, 24ull * n, hipMemcpyHostToDevice ) );
    BVH_TRY( hipEventRecord( t0, stream ) );
    BVH_TRY( hipMemsetAsync( dCounters.p, 0, 12, stream ) );
    FillItemsKernel<<< itemBlocks, kBlock, 0, stream >>>( dBoxes.As< float >(), dItems.As< float4 >(), n, dCounters.As< uint32_t >() );
    {
        uint32_t notFinite = 0;
        BVH_TRY( hipMemcpy( &notFinite, dCounters.p, 4, hipMemcpyDeviceToHost ) );
        if ( notFinite ) return Fail( PT_ERR_INVALID_SCENE, "a box has a bound that is not finite, or bounds whose sum is not (pass --host-bvh to the command line program to build such a tree on the host)" );
    }
    BVH_TRY( hipMemsetAsync( dSeg.p, 0, 4ull * n, stream ) );

    uint32_t counts[3] = { 0, 0, 0 }; // ranges this level splits; small ranges that the level before left to single threads; ranges without a box yet
    if ( n > kThreadRange )
    {
        BigRange root{};
        root.start = 0; root.end = n; root.slot = 0; root.cutoff = 0;
        root.left = root.right = kNone;
        root.needReduce = 1;
        for ( int q = 0; q < 12; ++q ) root.acc[q] = q < 6 ? 0xFFFFFFFFu : 0u;
        BVH_TRY( hipMemcpyAsync( dRanges[0].p, &root, sizeof( root ), hipMemcpyHostToDevice, stream ) );
        BVH_TRY( hipStreamSynchronize( stream ) ); // `root` leaves scope
        counts[0] = 1;
        counts[2] = 1;
    }
    else
    {
        const SmallRange root{ 0, n, 0 };
        BVH_TRY( hipMemcpyAsync( dSmall.p, &root, sizeof( root ), hipMemcpyHostToDevice, stream ) );
        BVH_TRY( hipStreamSynchronize( stream ) );
        counts[1] = 1;
    }

    int cur = 0;
    for ( uint32_t level = 0; counts[0] || counts[1]; ++level )
    {
        if ( level >= kMaxLevels ) return Fail( PT_ERR_INVALID_SCENE, "the tree is deeper than 8192 levels" );
        if ( counts[1] )
            SubtreeKernel<<< ( counts[1] + 63u ) / 64u, 64, 0, stream >>>( dItems.As< float4 >(), dSmall.As< SmallRange >(), counts[1], dNodes.As< uint32_t >(),
                                                                               dOrder.As< uint32_t >() );
        if ( !counts[0] ) break;
        BigRange* now        = dRanges[cur].As< BigRange >();
        BigRange* next       = dRanges[cur ^ 1].As< BigRange >();
        const uint32_t rangeBlocks = ( counts[0] + 255u ) / 256u;
        const uint32_t bucketWords = counts[0] * kBuckets * kBucketWords;
        InitBucketsKernel<<< ( bucketWords + 255u ) / 256u, 256, 0, stream >>>( dBuckets.As< uint32_t >(), bucketWords );
        int updateSeg = level > 0;
        if ( counts[2] ) // the root, or halves of a range that was halved where it lay: their boxes are summed from the shapes
        {
            if ( updateSeg ) SegUpdateKernel<<< itemBlocks, kBlock, 0, stream >>>( dSeg.As< uint32_t >(), next, n );
            updateSeg = 0;
            ReduceKernel<<< itemBlocks, kBlock, 0, stream >>>( dItems.As< float4 >(), dSeg.As< uint32_t >(), now, n );
            AxisKernel<<< rangeBlocks, 256, 0, stream >>>( now, counts[0], dNodes.As< uint32_t >() );
        }
        BucketKernel<<< bucketBlocks, kBlock, 0, stream >>>( dItems.As< float4 >(), dSeg.As< uint32_t >(), next, now, dBuckets.As< Buckets >(), n, tiles, updateSeg );
        BVH_TRY( hipMemsetAsync( dCounters.p, 0, 12, stream ) );
        SplitKernel<<< rangeBlocks, 256, 0, stream >>>( now, dBuckets.As< Buckets >(), counts[0], dNodes.As< uint32_t >(), next, dSmall.As< SmallRange >(),
                                                         dCounters.As< uint32_t >() );
        FlagKernel<<< padded / kBlock, kBlock, 0, stream >>>( dItems.As< float4 >(), dSeg.As< uint32_t >(), now, dPrefix.As< uint32_t >(), n, padded );
        ScanSumsKernel<<< scanBlocks, kBlock, 0, stream >>>( dPrefix.As< uint4 >(), dSums.As< uint32_t >() );
        ScanTopKernel<<< 1, kBlock, 0, stream >>>( dSums.As< uint32_t >(), scanBlocks );
        ScanApplyKernel<<< scanBlocks, kBlock, 0, stream >>>( dPrefix.As< uint4 >(), dSums.As< uint32_t >() );
        PartnerKernel<<< itemBlocks, kBlock, 0, stream >>>( dSeg.As< uint32_t >(), now, dPrefix.As< uint32_t >(), dFailing.As< uint32_t >(), dPassing.As< uint32_t >(), n );
        SwapKernel<<< itemBlocks, kBlock, 0, stream >>>( dItems.As< float4 >(), dSeg.As< uint32_t >(), now, dPrefix.As< uint32_t >(), dFailing.As< uint32_t >(),
                                                          dPassing.As< uint32_t >(), n );
        BVH_TRY( hipGetLastError() );
        BVH_TRY( hipMemcpy( counts, dCounters.p, 12, hipMemcpyDeviceToHost ) );
        if ( counts[0] > bigMax || counts[1] > n ) return Fail( PT_ERR_DEVICE, "range lists overflowed" );
        cur ^= 1;
    }
    BVH_TRY( hipEventRecord( t1, stream ) );
    BVH_TRY( hipMemcpy( nodesOut, dNodes.p, 32ull * ( 2ull * n - 1ull ), hipMemcpyDeviceToHost ) );
    BVH_TRY( hipMemcpy( orderOut, dOrder.p, 4ull * n, hipMemcpyDeviceToHost ) );
    BVH_TRY( hipEventSynchronize( t1 ) );
    if ( msOut ) BVH_TRY( hipEventElapsedTime( msOut, t0, t1 ) );
    if ( numNodesOut ) *numNodesOut = 2ull * n - 1ull;
    return PT_OK;
}

} // namespace

extern "C" int pt_build_bvh( int32_t device, const float* boxes, uint64_t n, PtBvhNode* nodes_out, uint32_t* order_out, uint64_t* num_nodes_out,
                             float* device_ms_out )
try
{
    static_assert( sizeof( PtBvhNode ) == 32, "PtBvhNode is the reference's 32-byte node" );
    return Build( device, boxes, n, nodes_out, order_out, num_nodes_out, device_ms_out );
}
catch ( ... )
{
    return ptd::SetError( PT_ERR_DEVICE, "pt_build_bvh: out of host memory" );
}
