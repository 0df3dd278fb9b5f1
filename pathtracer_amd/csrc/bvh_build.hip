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
//   * ranges of at most four shapes are finished by one thread each with the element moves of libstdc++'s
//     std::nth_element (GCC 11: one median-of-three partition step for four elements, then insertion sort).
// All arithmetic is float32 in the reference's operation order, compiled without contraction.
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
constexpr int kBlock           = 1024;
constexpr int kWaves           = kBlock / 64;
constexpr int kMaxRuns         = 32; // ranges a block may meet and still combine their sums in LDS
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

// A range of more than four shapes that this level splits.
struct BigRange
{
    uint32_t start, end, slot, cutoff;
    uint32_t left, right; // the ranges of the next level that the two halves become, or kNone
    uint32_t dim, noMove;
    uint32_t acc[12]; // Enc of: box lo xyz, centroid box lo xyz (minima), box hi xyz, centroid box hi xyz (maxima)
    float cMin, cMax; // centroid box along dim
    float area;       // SurfaceArea of the range's box
    uint32_t pad;
};
static_assert( sizeof( BigRange ) == 96, "BigRange is 24 words" );

struct SmallRange
{
    uint32_t start, end, slot;
};

// bucket sums of one range: [bucket][0] = count, [1..3] = Enc( lo ), [4..6] = Enc( hi )
struct Buckets
{
    uint32_t w[kBuckets][8];
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

// Level step 1: every shape finds the range it belongs to now (its old range's left or right half) and adds its
// box and centroid to that range's sums.
__global__ __launch_bounds__( kBlock ) void ReduceKernel( const float4* __restrict__ items, uint32_t* __restrict__ seg, const BigRange* __restrict__ prev,
                                                           BigRange* __restrict__ cur, uint32_t n, int level )
{
    __shared__ uint32_t sLast[kWaves], sHeads[kWaves];
    __shared__ uint32_t sAcc[kMaxRuns][12];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    uint32_t r       = kNone;
    if ( i < n )
    {
        r = seg[i];
        if ( level > 0 && r != kNone )
        {
            const BigRange* p = prev + r;
            r                 = i < p->cutoff ? p->left : p->right;
            seg[i]            = r;
        }
    }
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

// Level step 2: box, split axis and node record of every range; its bucket sums start empty.
__global__ void AxisKernel( BigRange* __restrict__ cur, Buckets* __restrict__ buckets, uint32_t count, uint32_t* __restrict__ nodes )
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if ( r >= count ) return;
    BigRange& R = cur[r];
    Box box, cbox;
    for ( int k = 0; k < 3; ++k )
    {
        box.lo[k]  = Dec( R.acc[k] );
        cbox.lo[k] = Dec( R.acc[3 + k] );
        box.hi[k]  = Dec( R.acc[6 + k] );
        cbox.hi[k] = Dec( R.acc[9 + k] );
    }
    const int dim = LongestDimension( cbox ); // src/bvh.cpp:68-70
    R.dim         = static_cast< uint32_t >( dim );
    R.cMin        = cbox.lo[dim];
    R.cMax        = cbox.hi[dim];
    R.area        = SurfaceArea( box );
    WriteNode( nodes, R.slot, box, 0u, 0u, static_cast< uint32_t >( dim ) ); // the second child's slot follows in SplitKernel
    for ( int b = 0; b < kBuckets; ++b )
    {
        uint4* w = reinterpret_cast< uint4* >( buckets[r].w[b] );
        w[0]     = uint4{ 0u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu };
        w[1]     = uint4{ 0u, 0u, 0u, 0u };
    }
}

// Level step 3: bucket counts and bucket boxes (src/bvh.cpp:104-112).
__global__ __launch_bounds__( kBlock ) void BucketKernel( const float4* __restrict__ items, const uint32_t* __restrict__ seg, const BigRange* __restrict__ cur,
                                                           Buckets* __restrict__ buckets, uint32_t n )
{
    __shared__ uint32_t sLast[kWaves], sHeads[kWaves];
    __shared__ uint32_t sBk[kMaxRuns][kBuckets][7];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t r = i < n ? seg[i] : kNone;
    uint32_t v[6];
    int b = 0;
    if ( r != kNone )
    {
        const float4 lo = items[2ull * i], hi = items[2ull * i + 1];
        const BigRange* R = cur + r;
        const uint32_t dim = R->dim;
        const float c = dim == 0 ? Centroid( lo.x, hi.x ) : dim == 1 ? Centroid( lo.y, hi.y ) : Centroid( lo.z, hi.z );
        b    = BucketOf( c, R->cMin, R->cMax );
        v[0] = Enc( lo.x ); v[1] = Enc( lo.y ); v[2] = Enc( lo.z );
        v[3] = Enc( hi.x ); v[4] = Enc( hi.y ); v[5] = Enc( hi.z );
    }
    uint32_t runs;
    bool head;
    const uint32_t run = RunIndex( r, sLast, sHeads, &runs, &head );
    if ( runs <= kMaxRuns )
    {
        for ( uint32_t k = threadIdx.x; k < runs * kBuckets * 7; k += kBlock )
        {
            const uint32_t word = k % 7u;
            ( &sBk[0][0][0] )[k] = word == 0 ? 0u : word < 4 ? 0xFFFFFFFFu : 0u;
        }
        __syncthreads();
        if ( r != kNone )
        {
            atomicAdd( &sBk[run][b][0], 1u );
            for ( int k = 0; k < 3; ++k ) atomicMin( &sBk[run][b][1 + k], v[k] );
            for ( int k = 3; k < 6; ++k ) atomicMax( &sBk[run][b][1 + k], v[k] );
        }
        __syncthreads();
        if ( head && r != kNone )
        {
            for ( int q = 0; q < kBuckets; ++q )
            {
                const uint32_t cnt = sBk[run][q][0];
                if ( cnt == 0 ) continue;
                uint32_t* g = buckets[r].w[q];
                atomicAdd( g, cnt );
                for ( int k = 0; k < 3; ++k ) atomicMin( g + 1 + k, sBk[run][q][1 + k] );
                for ( int k = 3; k < 6; ++k ) atomicMax( g + 1 + k, sBk[run][q][1 + k] );
            }
        }
    }
    else if ( r != kNone )
    {
        uint32_t* g = buckets[r].w[b];
        atomicAdd( g, 1u );
        for ( int k = 0; k < 3; ++k ) atomicMin( g + 1 + k, v[k] );
        for ( int k = 3; k < 6; ++k ) atomicMax( g + 1 + k, v[k] );
    }
}

__device__ __forceinline__ void NewRange( uint32_t start, uint32_t end, uint32_t slot, BigRange* next, uint32_t* counters, SmallRange* small, uint32_t* id )
{
    if ( end - start > kSmallRange )
    {
        const uint32_t k = atomicAdd( counters + 0, 1u );
        BigRange& R      = next[k];
        R.start = start; R.end = end; R.slot = slot; R.cutoff = start;
        R.left = R.right = kNone;
        R.dim = 0; R.noMove = 0;
        for ( int q = 0; q < 12; ++q ) R.acc[q] = q < 6 ? 0xFFFFFFFFu : 0u;
        *id = k;
    }
    else
    {
        const uint32_t k = atomicAdd( counters + 1, 1u );
        small[k]         = SmallRange{ start, end, slot };
        *id              = kNone;
    }
}

// Level step 4: the eleven split costs, the cheapest split, the partition point and the two ranges of the next
// level (src/bvh.cpp:114-169).
__global__ void SplitKernel( BigRange* __restrict__ cur, const Buckets* __restrict__ buckets, uint32_t count, uint32_t* __restrict__ nodes, BigRange* __restrict__ next,
                             SmallRange* __restrict__ small, uint32_t* __restrict__ counters )
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if ( r >= count ) return;
    BigRange& R = cur[r];
    Box bb[kBuckets];
    int cnt[kBuckets];
    for ( int b = 0; b < kBuckets; ++b )
    {
        const uint32_t* w = buckets[r].w[b];
        cnt[b]            = static_cast< int >( w[0] );
        // an empty bucket is the empty box: Dec( 0xFFFFFFFF ) is a NaN, so spell it out
        for ( int k = 0; k < 3; ++k )
        {
            bb[b].lo[k] = cnt[b] ? Dec( w[1 + k] ) : kFltMax;
            bb[b].hi[k] = cnt[b] ? Dec( w[4 + k] ) : -kFltMax;
        }
    }
    float minCost = 0;
    int minBucket = 0;
    for ( int i = 0; i < kBuckets - 1; ++i )
    {
        Box b0 = EmptyBox(), b1 = EmptyBox();
        int count0 = 0, count1 = 0;
        for ( int j = 0; j <= i; ++j )
        {
            Union( b0, bb[j] );
            count0 += cnt[j];
        }
        for ( int j = i + 1; j < kBuckets; ++j )
        {
            Union( b1, bb[j] );
            count1 += cnt[j];
        }
        const float cost = 0.5f + ( count0 * SurfaceArea( b0 ) + count1 * SurfaceArea( b1 ) ) / R.area;
        if ( i == 0 || cost < minCost )
        {
            minCost   = cost;
            minBucket = i;
        }
    }
    uint32_t passing = 0;
    for ( int j = 0; j <= minBucket; ++j ) passing += static_cast< uint32_t >( cnt[j] );
    uint32_t cutoff = R.start + passing;
    R.dim |= static_cast< uint32_t >( minBucket ) << 8;
    if ( cutoff == R.start || cutoff == R.end )
    {
        // every shape on one side: std::partition moved nothing; the range is halved where it lies (the reference
        // itself recurses without end here, src/bvh.cpp:173-175)
        cutoff   = static_cast< uint32_t >( ( static_cast< uint64_t >( R.start ) + R.end ) / 2u );
        R.noMove = 1;
    }
    R.cutoff              = cutoff;
    const uint32_t second = R.slot + 2u * ( cutoff - R.start );
    nodes[8ull * R.slot + 6] = second;
    NewRange( R.start, cutoff, R.slot + 1u, next, counters, small, &R.left );
    NewRange( cutoff, R.end, second, next, counters, small, &R.right );
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

// --- ranges of at most four shapes: one thread finishes the subtree -------------------------------------------------

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

__global__ void SmallKernel( const float4* __restrict__ items, const SmallRange* __restrict__ small, uint32_t count, uint32_t* __restrict__ nodes,
                             uint32_t* __restrict__ order )
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if ( r >= count ) return;
    const SmallRange R = small[r];
    const int total    = static_cast< int >( R.end - R.start );
    Shape v[kSmallRange];
    for ( int k = 0; k < total; ++k )
    {
        const float4 lo = items[2ull * ( R.start + k )], hi = items[2ull * ( R.start + k ) + 1];
        v[k].box   = Box{ { lo.x, lo.y, lo.z }, { hi.x, hi.y, hi.z } };
        v[k].index = __float_as_uint( lo.w );
    }
    int stackS[4], stackE[4];
    uint32_t stackSlot[4];
    int top      = 0;
    stackS[0]    = 0;
    stackE[0]    = total;
    stackSlot[0] = R.slot;
    top          = 1;
    while ( top > 0 )
    {
        --top;
        const int s = stackS[top], e = stackE[top];
        const uint32_t slot = stackSlot[top];
        Box box = EmptyBox();
        for ( int k = s; k < e; ++k ) Union( box, v[k].box );
        if ( e - s == 1 ) // src/bvh.cpp:47-56
        {
            WriteNode( nodes, slot, box, R.start + s, 1u, 0u );
            order[R.start + s] = v[s].index;
            continue;
        }
        Box cbox = EmptyBox();
        for ( int k = s; k < e; ++k )
        {
            Box c;
            for ( int q = 0; q < 3; ++q ) c.lo[q] = c.hi[q] = Centroid( v[k].box.lo[q], v[k].box.hi[q] );
            Union( cbox, c );
        }
        const int dim = LongestDimension( cbox );
        const int nth = ( s + e ) / 2; // src/bvh.cpp:85-90,95-98: the middle element, counted from the array's start
        NthElement( v, s, nth, e, dim );
        const uint32_t second = slot + 2u * static_cast< uint32_t >( nth - s );
        WriteNode( nodes, slot, box, second, 0u, static_cast< uint32_t >( dim ) );
        stackS[top] = nth; stackE[top] = e; stackSlot[top] = second; ++top;
        stackS[top] = s;   stackE[top] = nth; stackSlot[top] = slot + 1u; ++top;
    }
}

__global__ void FillItemsKernel( const float* __restrict__ boxes, float4* __restrict__ items, uint32_t n )
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    const float* b   = boxes + 6ull * i;
    items[2ull * i]     = float4{ b[0], b[1], b[2], __uint_as_float( i ) };
    items[2ull * i + 1] = float4{ b[3], b[4], b[5], 0.0f };
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
    for ( uint64_t k = 0; k < 6 * n64; ++k )
        if ( !std::isfinite( boxes[k] ) ) return Fail( PT_ERR_INVALID_SCENE, "a box has a bound that is not finite" );
    const uint32_t n      = static_cast< uint32_t >( n64 );
    const uint32_t padded = ( ( n + 1u + 4u * kBlock - 1u ) / ( 4u * kBlock ) ) * ( 4u * kBlock ); // prefix sums: n + 1 words, whole blocks of uint4
    const uint32_t scanBlocks = padded / ( 4u * kBlock );
    const uint32_t bigMax     = n / ( kSmallRange + 1u ) + 1u;
    const uint32_t itemBlocks = ( n + kBlock - 1u ) / kBlock;

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
    BVH_TRY( dCounters.Alloc( 8 ) );
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
    FillItemsKernel<<< itemBlocks, kBlock, 0, stream >>>( dBoxes.As< float >(), dItems.As< float4 >(), n );
    BVH_TRY( hipMemsetAsync( dSeg.p, 0, 4ull * n, stream ) );

    uint32_t counts[2] = { 0, 0 }; // ranges of more than four shapes of the level, ranges of at most four from the level before
    if ( n > kSmallRange )
    {
        BigRange root{};
        root.start = 0; root.end = n; root.slot = 0; root.cutoff = 0;
        root.left = root.right = kNone;
        for ( int q = 0; q < 12; ++q ) root.acc[q] = q < 6 ? 0xFFFFFFFFu : 0u;
        BVH_TRY( hipMemcpyAsync( dRanges[0].p, &root, sizeof( root ), hipMemcpyHostToDevice, stream ) );
        BVH_TRY( hipStreamSynchronize( stream ) ); // `root` leaves scope
        counts[0] = 1;
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
            SmallKernel<<< ( counts[1] + 255u ) / 256u, 256, 0, stream >>>( dItems.As< float4 >(), dSmall.As< SmallRange >(), counts[1], dNodes.As< uint32_t >(),
                                                                               dOrder.As< uint32_t >() );
        if ( !counts[0] ) break;
        BigRange* now        = dRanges[cur].As< BigRange >();
        BigRange* next       = dRanges[cur ^ 1].As< BigRange >();
        const uint32_t rangeBlocks = ( counts[0] + 255u ) / 256u;
        ReduceKernel<<< itemBlocks, kBlock, 0, stream >>>( dItems.As< float4 >(), dSeg.As< uint32_t >(), next, now, n, static_cast< int >( level ) );
        AxisKernel<<< rangeBlocks, 256, 0, stream >>>( now, dBuckets.As< Buckets >(), counts[0], dNodes.As< uint32_t >() );
        BucketKernel<<< itemBlocks, kBlock, 0, stream >>>( dItems.As< float4 >(), dSeg.As< uint32_t >(), now, dBuckets.As< Buckets >(), n );
        BVH_TRY( hipMemsetAsync( dCounters.p, 0, 8, stream ) );
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
        BVH_TRY( hipMemcpy( counts, dCounters.p, 8, hipMemcpyDeviceToHost ) );
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
