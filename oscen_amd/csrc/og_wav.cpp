// og_wav.cpp -- the output step immediately downstream of the path: interleaved bus -> RIFF/WAVE
// file (the reference examples use the `hound` crate for this: 16-bit PCM or 32-bit IEEE float).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/oscen_gpu.h"
#include "og_abi.h"

namespace {
void put_u32(std::vector<uint8_t>& b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
void put_u16(std::vector<uint8_t>& b, uint16_t v) { b.push_back((uint8_t)v); b.push_back((uint8_t)(v >> 8)); }
} // namespace

extern "C" {
int og_write_wav(const char* path, const float* interleaved, uint64_t frames, uint32_t channels,
                 uint32_t sample_rate, uint32_t bits_per_sample)
{
    return ogabi::guard([&]() -> int { // (the file image is assembled in a vector: bad_alloc -> OG_E_NOMEM)
    if (!path || (!interleaved && frames) || channels == 0 || (bits_per_sample != 16 && bits_per_sample != 32))
        return OG_E_INVALID;
    const bool f32 = bits_per_sample == 32;
    const uint64_t n = frames * channels;
    const uint64_t data_bytes = n * (bits_per_sample / 8);
    if (data_bytes > 0xFFFFFFFFull - 44) return OG_E_INVALID;
    std::vector<uint8_t> b;
    b.reserve(44 + (size_t)data_bytes);
    b.insert(b.end(), {'R', 'I', 'F', 'F'});
    put_u32(b, (uint32_t)(36 + data_bytes));
    b.insert(b.end(), {'W', 'A', 'V', 'E', 'f', 'm', 't', ' '});
    put_u32(b, 16);
    put_u16(b, f32 ? 3 : 1); // WAVE_FORMAT_IEEE_FLOAT / WAVE_FORMAT_PCM
    put_u16(b, (uint16_t)channels);
    put_u32(b, sample_rate);
    put_u32(b, sample_rate * channels * (bits_per_sample / 8));
    put_u16(b, (uint16_t)(channels * (bits_per_sample / 8)));
    put_u16(b, (uint16_t)bits_per_sample);
    b.insert(b.end(), {'d', 'a', 't', 'a'});
    put_u32(b, (uint32_t)data_bytes);
    for (uint64_t i = 0; i < n; ++i) {
        const float x = interleaved[i];
        if (f32) {
            uint32_t u;
            memcpy(&u, &x, 4);
            put_u32(b, u);
        } else {
            float c = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);
            if (c != c) c = 0.0f;
            put_u16(b, (uint16_t)(int16_t)lrintf(c * 32767.0f));
        }
    }
    FILE* f = fopen(path, "wb");
    if (!f) return OG_E_INVALID;
    const bool ok = fwrite(b.data(), 1, b.size(), f) == b.size();
    fclose(f);
    return ok ? OG_OK : OG_E_INVALID;
    });
}
} // extern "C"
