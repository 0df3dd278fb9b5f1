/*
 * pt_rng.h - the deterministic replacement for the reference's Random::Rand(), for the CPU checker and the
 * reference harness.  TEST INFRASTRUCTURE: nothing under oracle/ is part of the shipped product.
 *
 * The reference draws from a thread_local std::mt19937 seeded by std::random_device
 * (src/utils/random.cpp:9-21), so two runs never agree.  Every sample consumes its draws in a fixed
 * sequence (draw 0 and 1 for the jitter, then a constant number per bounce), so a stateless generator
 * indexed by (pixel, sample, draw) reproduces a sequential stream per sample:
 *
 *   PT_RNG_PHILOX: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11) with
 *                  key = (seed low, seed high), counter = (pixel, sample, draw / 4, 0); the draw is word
 *                  draw % 4 of the block, mapped to [0,1) as (word >> 8) * 2^-24.
 *   PT_RNG_TABLE:  table[((pixel * spp_total + sample) * draws_per_sample + draw) % table_len].
 */
#ifndef PT_RNG_H
#define PT_RNG_H

#include <stdint.h>

static inline void pt_philox4x32_10( const uint32_t counter[4], const uint32_t key[2], uint32_t out[4] )
{
    uint32_t c0 = counter[0], c1 = counter[1], c2 = counter[2], c3 = counter[3];
    uint32_t k0 = key[0], k1 = key[1];
    for ( int round = 0; round < 10; ++round )
    {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)( p1 >> 32 ) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)( p0 >> 32 ) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct PtRngState
{
    uint32_t mode;            /* PT_RNG_PHILOX = 0, PT_RNG_TABLE = 1 */
    uint64_t seed;
    const float* table;
    uint64_t table_len;
    int32_t  spp_total;
    int32_t  draws_per_sample;
    uint32_t pixel, sample, draw; /* draw = next draw index of the current sample */
} PtRngState;

static inline float pt_rng_draw( const PtRngState* s, uint32_t draw )
{
    if ( s->mode == 1 )
    {
        uint64_t i = ( (uint64_t)s->pixel * (uint64_t)s->spp_total + s->sample ) * (uint64_t)s->draws_per_sample + draw;
        return s->table[i % s->table_len];
    }
    uint32_t counter[4] = { s->pixel, s->sample, draw >> 2, 0u };
    uint32_t key[2]     = { (uint32_t)s->seed, (uint32_t)( s->seed >> 32 ) };
    uint32_t out[4];
    pt_philox4x32_10( counter, key, out );
    return (float)( out[draw & 3u] >> 8 ) * ( 1.0f / 16777216.0f );
}

static inline float pt_rng_next( PtRngState* s ) { return pt_rng_draw( s, s->draw++ ); }

#endif /* PT_RNG_H */
