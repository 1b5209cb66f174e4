// tests/emu/inflate_check.cpp -- TEST-ONLY: the two DEFLATE decoders of the device ingest against zlib: inflate_core.hpp (one wavefront per block, stepped with one lane) and
// inflate_fast_core.hpp (round 5: pass 1 a lane per block, pass 2 the noted matches 64 at a time -- the rounds of bgzf_inflate_resolve_kernel are stepped here with 64 lanes whose
// loads of a round all come before its stores) on every kind of block (stored, fixed, dynamic), runs that overlap themselves, matches from far back, damaged streams (an error or a
// wrong size, never a write outside the buffers), and hand-made dynamic headers with incomplete / over-subscribed / single-symbol / empty codes, where the verdict must be zlib's.
// Built and run by tests/test_host_and_device_logic.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <algorithm>
#include <utility>
#include <zlib.h>
#include "../../arriba_amd/csrc/device/inflate_fast_core.hpp"
using namespace agpu;
static const size_t PAD = 256; // what the readers of the DEFLATE stream may read behind its end (inflate_fast_core.hpp: FastBits)
static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t>& in, int level, int strategy) {
	z_stream z; memset(&z, 0, sizeof(z));
	deflateInit2(&z, level, Z_DEFLATED, -15, 8, strategy);
	std::vector<uint8_t> out(in.size() * 2 + 1024);
	z.next_in = (Bytef*) in.data(); z.avail_in = in.size(); z.next_out = out.data(); z.avail_out = out.size();
	int rc = deflate(&z, Z_FINISH); if (rc != Z_STREAM_END) { printf("deflate failed\n"); exit(1); }
	out.resize(z.total_out); deflateEnd(&z); return out;
}
// pass 2 as the device runs it: groups of 64 notes, rounds; within a round every ready lane reads (all lanes first), then every ready lane writes
static void resolve_in_rounds(uint8_t* out, const unsigned long long* notes, uint32_t n, unsigned long long& rounds) {
	for (uint32_t base = 0; base < n; base += 64) {
		const uint32_t lanes = n - base < 64 ? n - base : 64;
		bool pending[64];
		for (uint32_t l = 0; l < 64; ++l) pending[l] = l < lanes;
		while (true) {
			uint32_t first = 64;
			for (uint32_t l = 0; l < lanes; ++l) if (pending[l]) { first = l; break; }
			if (first == 64) break;
			++rounds;
			const uint32_t frontier = inflate_note_position(notes[base + first]);
			std::vector<std::vector<uint8_t>> copies(lanes);
			for (uint32_t l = 0; l < lanes; ++l) { // loads
				const unsigned long long note = notes[base + l];
				const uint32_t position = inflate_note_position(note), length = inflate_note_length(note), distance = inflate_note_distance(note);
				if (!pending[l] || !inflate_match_is_ready(position, length, distance, frontier)) continue;
				std::vector<uint8_t> scratch(out, out + position + length + 16); // the lane's own copy runs on a snapshot: it sees nothing another lane writes in this round
				inflate_copy_match(scratch.data(), position, length, distance);
				copies[l].assign(scratch.begin() + position, scratch.begin() + position + length);
			}
			for (uint32_t l = 0; l < lanes; ++l) if (!copies[l].empty()) { memcpy(out + inflate_note_position(notes[base + l]), copies[l].data(), copies[l].size()); pending[l] = false; }
		}
	}
}
static InflateFastTables* g_tables = new InflateFastTables();
static std::vector<unsigned long long> g_notes(INFLATE_MATCH_CAPACITY + 8, 0x5555555555555555ull);
// both passes; returns the status; `out` has `size` bytes + 64 of 0xCD behind them
static int fast_inflate(const std::vector<uint8_t>& packed /* + PAD bytes of padding */, std::vector<uint8_t>& out, uint32_t size, bool in_rounds, unsigned long long& rounds, uint32_t* notes_used = nullptr) {
	uint32_t n = 0;
	const int rc = inflate_tokens(packed.data(), (uint32_t) packed.size() - PAD, out.data(), size, g_notes.data(), INFLATE_MATCH_CAPACITY, n, *g_tables);
	if (g_notes[INFLATE_MATCH_CAPACITY] != 0x5555555555555555ull) { printf("NOTES OVERRUN\n"); exit(1); }
	if (notes_used) *notes_used = n;
	if (rc != INFLATE_OK) return rc;
	if (in_rounds) resolve_in_rounds(out.data(), g_notes.data(), n, rounds); else inflate_resolve_in_order(out.data(), g_notes.data(), n);
	return rc;
}
static int zlib_inflate_raw(const uint8_t* in, size_t in_size, std::vector<uint8_t>& out) {
	z_stream z; memset(&z, 0, sizeof(z));
	inflateInit2(&z, -15);
	out.assign(70000, 0);
	z.next_in = (Bytef*) in; z.avail_in = in_size; z.next_out = out.data(); z.avail_out = out.size();
	const int rc = inflate(&z, Z_FINISH);
	out.resize(z.total_out); inflateEnd(&z);
	return rc == Z_STREAM_END ? 0 : 1;
}
// a dynamic block written by hand: the code-length code gives every length 0..15 a 4-bit code (complete), so any set of lengths can be written down -- also those a compressor never makes
struct BitWriter { std::vector<uint8_t> bytes; uint32_t bit = 0;
	void put(uint32_t value, int n) { for (int k = 0; k < n; ++k) { if (bit == 0) bytes.push_back(0); bytes.back() |= ((value >> k) & 1u) << bit; bit = (bit + 1) & 7; } }
	void put_code(uint32_t code, int length) { for (int k = length - 1; k >= 0; --k) put((code >> k) & 1u, 1); } };
static void canonical(const std::vector<int>& lengths, std::vector<uint32_t>& codes) {
	int count[16] = { 0 }; for (int l : lengths) count[l]++; count[0] = 0;
	uint32_t next[16] = { 0 }, code = 0; for (int l = 1; l < 16; ++l) { code = (code + count[l - 1]) << 1; next[l] = code; }
	codes.assign(lengths.size(), 0); for (size_t s = 0; s < lengths.size(); ++s) if (lengths[s]) codes[s] = next[lengths[s]]++;
}
static std::vector<uint8_t> handmade_block(const std::vector<int>& litlen, const std::vector<int>& distance, const std::vector<std::pair<int, int>>& tokens /* (literal, -1) or (length symbol, distance symbol), extra bits 0 */) {
	BitWriter w;
	w.put(1, 1); w.put(2, 2); w.put((uint32_t) litlen.size() - 257, 5); w.put((uint32_t) distance.size() - 1, 5); w.put(19 - 4, 4);
	static const int order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
	for (int k = 0; k < 19; ++k) w.put(order[k] < 16 ? 4 : 0, 3);
	for (int l : litlen) w.put_code((uint32_t) l, 4);   // (lengths 0..15 all have 4 bits: the canonical code of length value v is v)
	for (int l : distance) w.put_code((uint32_t) l, 4);
	std::vector<uint32_t> litlen_codes, distance_codes; canonical(litlen, litlen_codes); canonical(distance, distance_codes);
	for (const std::pair<int, int>& token : tokens) {
		w.put_code(litlen_codes[token.first], litlen[token.first]);
		if (token.second >= 0) w.put_code(distance_codes[token.second], distance[token.second]);
	}
	w.put_code(litlen_codes[256], litlen[256]);
	return w.bytes;
}

int main() {
	std::mt19937 rng(7);
	InflateShared* shared = new InflateShared();
	int checked = 0, failures = 0;
	unsigned long long rounds = 0, groups = 0, retries = 0;
	for (int kind = 0; kind < 8; ++kind)
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
			case 7: data[i] = i < 64 ? (uint8_t) (rng() & 255) : (rng() % 5 == 0 ? (uint8_t) (rng() & 255) : data[i - 1 - rng() % (i < 12 ? i : 12)]); break; // chains of short matches that read each other's output (many rounds in pass 2), periods below 8
			}
		}
		std::vector<uint8_t> packed = deflate_raw(data, level, strategy);
		packed.resize(packed.size() + PAD, 0xAA); // the padding the kernel's buffer has
		std::vector<uint8_t> out(size + 64, 0xCD);
		auto sync = [] {}; auto broadcast = [](uint32_t v) { return v; };
		int rc = inflate_block(packed.data(), (uint32_t) packed.size() - PAD, out.data(), (uint32_t) size, *shared, 0, 1, sync, broadcast);
		++checked;
		bool ok = rc == INFLATE_OK && (size == 0 || memcmp(out.data(), data.data(), size) == 0) && out[size] == 0xCD;
		if (!ok) { ++failures; if (failures < 10) printf("FAIL kind %d size %d level %d strategy %d rc %d\n", kind, size, level, strategy, rc); }
		for (int in_rounds = 0; in_rounds < 2; ++in_rounds) { // the two passes of round 5
			std::fill(out.begin(), out.end(), 0xCD);
			uint32_t used = 0;
			rc = fast_inflate(packed, out, (uint32_t) size, in_rounds != 0, rounds, &used);
			if (in_rounds) groups += (used + 63) / 64;
			if (rc == INFLATE_RETRY) { if (in_rounds) ++retries; continue; } // (more matches than there is room to note: the device hands such a block to the other decoder)
			ok = rc == INFLATE_OK && (size == 0 || memcmp(out.data(), data.data(), size) == 0);
			for (int k = 0; k < 64; ++k) ok = ok && out[size + k] == 0xCD;
			if (!ok) { ++failures; if (failures < 10) printf("FAIL (two passes, %s) kind %d size %d level %d strategy %d rc %d\n", in_rounds ? "rounds" : "in order", kind, size, level, strategy, rc); }
		}
		// damaged streams must end with an error or a wrong size, never with a write outside the buffer
		if (size > 100) for (int trial = 0; trial < 3; ++trial) {
			std::vector<uint8_t> bad = packed; bad[rng() % (bad.size() - PAD)] ^= 1u << (rng() & 7);
			std::fill(out.begin(), out.end(), 0xCD);
			inflate_block(bad.data(), (uint32_t) bad.size() - PAD, out.data(), (uint32_t) size, *shared, 0, 1, sync, broadcast);
			if (out[size] != 0xCD) { ++failures; printf("OVERRUN kind %d size %d\n", kind, size); }
			std::fill(out.begin(), out.end(), 0xCD);
			const int verdict = fast_inflate(bad, out, (uint32_t) size, false, rounds);
			for (int k = 0; k < 64; ++k) if (out[size + k] != 0xCD) { ++failures; printf("OVERRUN (two passes) kind %d size %d\n", kind, size); break; }
			// (what zlib says of the same damaged stream: an error there must be an error here -- a stream zlib accepts may still be refused for its size)
			std::vector<uint8_t> reference;
			const int theirs = zlib_inflate_raw(bad.data(), bad.size() - PAD, reference);
			if (theirs == 0 && reference.size() == (size_t) size && verdict == INFLATE_OK && memcmp(out.data(), reference.data(), size) != 0) { ++failures; printf("DAMAGED STREAM DECODED DIFFERENTLY kind %d size %d\n", kind, size); }
			if (theirs != 0 && verdict == INFLATE_OK) { ++failures; printf("ACCEPTED WHAT ZLIB REFUSES kind %d size %d level %d strategy %d\n", kind, size, level, strategy); }
		}
	}
	// hand-made headers: the verdict on a set of code lengths must be zlib's (advisor, round 4: incomplete codes)
	int handmade = 0, accepted = 0, handed_back = 0;
	auto check_handmade = [&](const std::vector<int>& litlen, const std::vector<int>& distance, const std::vector<std::pair<int, int>>& tokens, const char* what) {
		std::vector<uint8_t> stream = handmade_block(litlen, distance, tokens);
		std::vector<uint8_t> reference;
		const int theirs = zlib_inflate_raw(stream.data(), stream.size(), reference);
		stream.resize(stream.size() + PAD, 0);
		std::vector<uint8_t> out(reference.size() + 64, 0xCD);
		int ours = fast_inflate(stream, out, (uint32_t) reference.size(), true, rounds);
		if (ours == INFLATE_RETRY) { // (second tables that a lane has no room for: the block goes to the other decoder, as on the device)
			++handed_back;
			auto sync = [] {}; auto broadcast = [](uint32_t v) { return v; };
			std::fill(out.begin(), out.end(), 0xCD);
			ours = inflate_block(stream.data(), (uint32_t) stream.size() - PAD, out.data(), (uint32_t) reference.size(), *shared, 0, 1, sync, broadcast);
		}
		++handmade;
		const bool same = theirs == 0 ? (ours == INFLATE_OK && memcmp(out.data(), reference.data(), reference.size()) == 0) : ours != INFLATE_OK;
		if (theirs == 0) ++accepted;
		if (!same) { ++failures; if (failures < 20) printf("HANDMADE %s: zlib %s, two passes rc %d\n", what, theirs == 0 ? "accepts" : "refuses", ours); }
	};
	{ std::vector<int> litlen(257, 0), distance(1, 0);
	  litlen[256] = 1; check_handmade(litlen, distance, {}, "a single end-of-block code of one bit, no distance code");
	  litlen['a'] = 1; check_handmade(litlen, distance, { { 'a', -1 }, { 'a', -1 } }, "two codes of one bit, no distance code");
	  litlen['a'] = 2; litlen[256] = 2; check_handmade(litlen, distance, { { 'a', -1 } }, "incomplete literal / length code");
	  litlen['b'] = 2; litlen['c'] = 2; litlen['d'] = 2; check_handmade(litlen, distance, { { 'a', -1 } }, "over-subscribed literal / length code");
	  litlen.assign(258, 0); litlen['a'] = 2; litlen['b'] = 2; litlen[256] = 2; litlen[257] = 2; // (257: length 3)
	  distance.assign(2, 0); distance[0] = 1; check_handmade(litlen, distance, { { 'a', -1 }, { 257, 0 } }, "a single distance code of one bit");
	  distance[0] = 2; check_handmade(litlen, distance, { { 'a', -1 }, { 257, 0 } }, "a single distance code of two bits (incomplete)");
	  distance[0] = 2; distance[1] = 2; check_handmade(litlen, distance, { { 'a', -1 }, { 257, 0 } }, "two distance codes of two bits (incomplete)");
	  distance[0] = 1; distance[1] = 1; check_handmade(litlen, distance, { { 'a', -1 }, { 'b', -1 }, { 257, 1 }, { 257, 0 } }, "two distance codes of one bit");
	  distance.assign(1, 0); check_handmade(litlen, distance, { { 'a', -1 }, { 257, 0 } }, "a match without any distance code");
	  check_handmade(litlen, distance, { { 'a', -1 }, { 'b', -1 } }, "no distance code, literals only");
	}
	for (int trial = 0; trial < 4000; ++trial) { // random sets of lengths, complete ones among them; literals and matches with codes of up to 15 bits (second-level tables, the long distance codes)
		const int n_litlen = 257 + rng() % 30, n_distance = 1 + rng() % 30;
		auto random_code = [&](int n, bool complete, int must_have) {
			std::vector<int> lengths(n, 0);
			if (!complete) { for (int s = 0; s < n; ++s) lengths[s] = rng() % 3 ? 0 : 1 + rng() % 15; if (must_have >= 0 && lengths[must_have] == 0) lengths[must_have] = 1 + rng() % 15; return lengths; }
			// a complete code: split leaves of a binary tree at random until enough symbols
			if (n < 2) { lengths[0] = rng() % 2; return lengths; }
			std::vector<int> leaves(1, 0);
			const int wanted = 2 + rng() % (n - 1);
			while ((int) leaves.size() < wanted) { const size_t k = rng() % leaves.size(); if (leaves[k] >= 15) { bool all = true; for (int l : leaves) all = all && l >= 15; if (all) break; continue; } leaves[k]++; leaves.push_back(leaves[k]); }
			std::vector<int> symbols(n); for (int s = 0; s < n; ++s) symbols[s] = s;
			std::shuffle(symbols.begin(), symbols.end(), rng);
			if (must_have >= 0) for (size_t k = 0; k < symbols.size(); ++k) if (symbols[k] == must_have) { std::swap(symbols[0], symbols[k]); break; }
			for (size_t k = 0; k < leaves.size(); ++k) lengths[symbols[k]] = leaves[k];
			return lengths;
		};
		const bool complete = rng() % 4 != 0;
		std::vector<int> litlen = random_code(n_litlen, complete, 256), distance = random_code(n_distance, complete || rng() % 2, -1);
		std::vector<std::pair<int, int>> tokens;
		uint32_t produced = 0;
		for (int k = 0; k < 200; ++k) {
			const int symbol = rng() % n_litlen;
			if (litlen[symbol] == 0 || symbol == 256) continue;
			if (symbol < 256) { tokens.push_back({ symbol, -1 }); ++produced; continue; }
			if (symbol > 264 || produced == 0) continue; // (length codes without extra bits; distance codes without extra bits: symbols 0..3)
			const int d = rng() % 4;
			if (d >= n_distance || distance[d] == 0 || (uint32_t) d + 1 > produced) continue;
			tokens.push_back({ symbol, d }); produced += 3 + (symbol - 257);
		}
		check_handmade(litlen, distance, tokens, "random code");
	}
	printf("%d blocks checked, %d failures; two passes: %llu groups of 64 matches in %llu rounds, %llu blocks handed back; %d hand-made headers (%d of them valid for zlib, %d handed to the other decoder)\n", checked, failures, groups, rounds, retries, handmade, accepted, handed_back);
	return failures != 0;
}
