// rt_film.h — device film (internal): ACES -> gamma -> u8 on the GPU, image.h:49-82.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace rt {
struct FilmTable {
    float thr[256];      // thr[k] = smallest ACES value whose quantised level is >= k (thr[0] = 0)
    uint32_t special[4]; // levels of NaN, negative finite, -inf (as the host film produces them); pad
};
// host/film.cpp: builds and verifies the table against the host film's own powf path; false = unusable
bool film_table(float thr[256], uint32_t special[3]);
// rt_film.hip: pixels of this shard (blocks of shard_block pixels, block b belongs to shard b % shard_count) -> rgb8
hipError_t launch_film(const float *fb_rgb, uint8_t *out_rgb8, uint32_t n_pixels, uint32_t shard_index, uint32_t shard_count, uint32_t shard_block,
                       const FilmTable *d_table, hipStream_t stream);
} // namespace rt
