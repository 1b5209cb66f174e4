// tests/emu/inflate_check.cpp -- TEST-ONLY: inflate_core.hpp (the decoder of bgzf_inflate_kernel) with one lane against zlib: every kind of block (stored, fixed, dynamic), runs that overlap
// themselves, matches from behind the ring, damaged streams (an error or a wrong size, never a write outside the buffer).  Built and run by tests/test_host_and_device_logic.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <zlib.h>
#include "../../arriba_amd/csrc/device/inflate_core.hpp"
using namespace agpu;
static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t>& in, int level, int strategy) {
	z_stream z; memset(&z, 0, sizeof(z));
	deflateInit2(&z, level, Z_DEFLATED, -15, 8, strategy);
	std::vector<uint8_t> out(in.size() * 2 + 1024);
	z.next_in = (Bytef*) in.data(); z.avail_in = in.size(); z.next_out = out.data(); z.avail_out = out.size();
	int rc = deflate(&z, Z_FINISH); if (rc != Z_STREAM_END) { printf("deflate failed\n"); exit(1); }
	out.resize(z.total_out); deflateEnd(&z); return out;
}
int main() {
	std::mt19937 rng(7);
	InflateShared* shared = new InflateShared();
	int checked = 0, failures = 0;
	for (int kind = 0; kind < 7; ++kind)
	for (int size : { 0, 1, 2, 100, 4097, 20000, 65280, 65536 })
	for (int level : { 0, 1, 6, 9 })
	for (int strategy : { Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED }) {
		std::vector<uint8_t> data(size);
		for (int i = 0; i < size; ++i) {
			switch (kind) {
			case 0: data[i] = rng() & 255; break;                         // incompressible
			case 1: data[i] = "ACGT"[rng() & 3]; break;                   // sequence-like
			case 2: data[i] = 'A'; break;                                  // one long run (overlapping matches, distance 1)
			case 3: data[i] = (uint8_t) (i % 251); break;                  // period 251
			case 4: data[i] = i < 40000 ? (uint8_t) (rng() & 255) : data[i - 33000 + (i % 7)]; break; // matches from far back (> 16 KB: behind the ring)
			case 5: data[i] = (rng() % 100 < 97) ? 'x' : (uint8_t) (rng() & 255); break; // skewed: long and short codes
			case 6: data[i] = (uint8_t) ((i * 2654435761u) >> 24 & 15); break;
			}
		}
		std::vector<uint8_t> packed = deflate_raw(data, level, strategy);
		packed.resize(packed.size() + 16, 0xAA); // the padding the kernel's buffer has
		std::vector<uint8_t> out(size + 64, 0xCD);
		auto sync = [] {}; auto broadcast = [](uint32_t v) { return v; };
		int rc = inflate_block(packed.data(), (uint32_t) packed.size() - 16, out.data(), (uint32_t) size, *shared, 0, 1, sync, broadcast);
		++checked;
		bool ok = rc == INFLATE_OK && memcmp(out.data(), data.data(), size) == 0 && out[size] == 0xCD;
		if (!ok) { ++failures; if (failures < 10) printf("FAIL kind %d size %d level %d strategy %d rc %d\n", kind, size, level, strategy, rc); }
		// damaged streams must end with an error or a wrong size, never with a write outside the buffer
		if (size > 100) for (int trial = 0; trial < 3; ++trial) {
			std::vector<uint8_t> bad = packed; bad[rng() % (bad.size() - 16)] ^= 1u << (rng() & 7);
			std::fill(out.begin(), out.end(), 0xCD);
			inflate_block(bad.data(), (uint32_t) bad.size() - 16, out.data(), (uint32_t) size, *shared, 0, 1, sync, broadcast);
			if (out[size] != 0xCD) { ++failures; printf("OVERRUN kind %d size %d\n", kind, size); }
		}
	}
	printf("%d blocks checked, %d failures\n", checked, failures);
	return failures != 0;
}
