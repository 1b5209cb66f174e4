// TEST-ONLY: the CRC-32 of a block as bgzf_crc_kernel computes it (crc32_core.hpp: the payload as the end of a virtual 64 KB block, raw CRCs of 64 chunks of 1 KB, folded with the two
// table-driven operators) against zlib's crc32 -- every length from 0 to 300, lengths around the chunk and word boundaries up to 64 KB, random and constant bytes, at every alignment of
// the payload in memory.  usage: crc_check -> "N payloads, 0 failures"
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include <zlib.h>
#include "../../arriba_amd/csrc/device/crc32_core.hpp"

using namespace agpu;

static uint32_t by_chunks(const Crc32Tables& t, const uint8_t* payload, uint32_t n) {
	if (n < 4) return crc32_of_sliced(t.slice, payload, n);
	uint32_t chunk[CRC32_WAVE_LANES];
	const uint32_t first_word = (CRC32_VIRTUAL - n) / 4;
	for (uint32_t lane = 0; lane < CRC32_WAVE_LANES; ++lane) {
		uint32_t c = 0;
		for (uint32_t w = lane * (CRC32_WAVE_CHUNK / 4); w < (lane + 1) * (CRC32_WAVE_CHUNK / 4); ++w) c = crc32_raw_step(t.slice, c, w >= first_word ? crc32_virtual_word(payload, n, w) : 0u);
		chunk[lane] = c;
	}
	return crc32_fold_chunks(t, chunk);
}

int main() {
	static Crc32Tables tables;
	crc32_make_tables(tables);
	std::mt19937 rng(20260927);
	std::vector<uint32_t> lengths;
	for (uint32_t n = 0; n <= 300; ++n) lengths.push_back(n);
	for (uint32_t base : { 1020u, 1024u, 2048u, 4096u, 8188u, 8192u, 16384u, 32768u, 65276u, 65280u, 65532u, 65536u }) for (int d = -5; d <= 5; ++d) if ((int64_t) base + d >= 0 && base + d <= CRC32_VIRTUAL) lengths.push_back(base + d);
	for (int k = 0; k < 300; ++k) lengths.push_back(rng() % (CRC32_VIRTUAL + 1));
	std::vector<uint8_t> memory(CRC32_VIRTUAL + 64);
	unsigned long long checked = 0, failures = 0;
	for (uint32_t n : lengths)
		for (uint32_t alignment = 0; alignment < 4; ++alignment)
			for (int kind = 0; kind < 3; ++kind) {
				uint8_t* payload = memory.data() + 16 + alignment;
				for (uint32_t i = 0; i < n; ++i) payload[i] = kind == 0 ? (uint8_t) rng() : kind == 1 ? 0 : 0xFF;
				payload[n] = 0xA5; // (a byte behind the payload must not matter)
				const uint32_t expected = (uint32_t) crc32(crc32(0L, Z_NULL, 0), payload, n);
				++checked;
				if (by_chunks(tables, payload, n) != expected) { if (++failures < 10) fprintf(stderr, "length %u, alignment %u, kind %d: differs from zlib\n", n, alignment, kind); }
			}
	printf("%llu payloads, %llu failures\n", checked, failures);
	return failures != 0;
}
