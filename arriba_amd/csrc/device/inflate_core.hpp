// inflate_core.hpp -- the DEFLATE stream of one BGZF block (RFC 1951; a BGZF block is a gzip member of at most 64 KB of data) decoded by one
// wavefront: the reference reads any BAM file htslib opens (source/read_chimeric_alignments.cpp:563-566), and what STAR writes by default is deflated.  With this the
// compressed bytes cross PCIe and the stream is made in HBM (agpu_ingest.hip: bgzf_inflate_kernel); the bytes are those zlib's inflate gives (tests: against zlib on
// every kind of block, stepped on the host with one lane).
//
// How the work is split among the lanes.  Huffman decoding is a chain of dependent steps (the length of a code is known only when it has been looked up), so one lane
// decodes: bits from a 64-bit buffer refilled with 4-byte loads issued a word ahead, symbols through a 10-bit (literals / lengths) and a 9-bit (distances) first-level table in LDS, the
// few longer codes bit by bit over the canonical counts (the method of zlib's puff.c).  Everything that is not a chain is done by all lanes: the tables are filled by all of them,
// a match (up to 258 bytes from up to 32 KB back, possibly overlapping itself) is copied by all of them at once -- byte i of the match is byte (i mod distance) of the
// bytes that stand at the distance, all of which are final when the match starts --, and the output goes through a ring of 8 KB in LDS that all lanes write back to
// HBM in runs of 4 KB, coalesced.  A match that reaches further back than the ring reads what was written back.  A short match from inside the ring is copied by the decoding lane itself.
//
// `lanes` = 64 on the device, 1 on the host (tests/emu, tests/test_inflate_core.py): every loop over "my share" degenerates to the sequential loop.
#ifndef AGPU_INFLATE_CORE_HPP
#define AGPU_INFLATE_CORE_HPP 1

#include "views.hpp"

namespace agpu {

const uint32_t INFLATE_RING = 8192, INFLATE_FLUSH = 4096, INFLATE_SHORT_MATCH = 24; // (a ring of 16 KB: 7 wavefronts per CU by their LDS; 8 KB: 12 -- the decoding lanes wait for LDS and HBM most of the time, so more of them in flight is what counts)
const int INFLATE_LITLEN_BITS = 10, INFLATE_DISTANCE_BITS = 9;
enum { INFLATE_OK = 0, INFLATE_BAD_BLOCK_TYPE = 1, INFLATE_BAD_STORED_LENGTH = 2, INFLATE_BAD_CODE_LENGTHS = 3, INFLATE_BAD_SYMBOL = 4, INFLATE_BAD_DISTANCE = 5, INFLATE_OUTPUT_OVERRUN = 6, INFLATE_INPUT_OVERRUN = 7,
       INFLATE_SIZE_MISMATCH = 8 };

struct InflateHuffman { // one code: first-level table + canonical counts for the codes that are longer than the table is wide
	uint16_t count[16];    // codes of every length
	uint16_t symbol[288];  // symbols ordered by (length, symbol)
};
struct InflateShared { // the memory the lanes of one wavefront share (LDS on the device)
	uint8_t ring[INFLATE_RING];
	uint16_t litlen_fast[1 << INFLATE_LITLEN_BITS], distance_fast[1 << INFLATE_DISTANCE_BITS]; // symbol << 4 | length of the code; 0: not a code of at most that many bits
	InflateHuffman litlen, distance;
	uint8_t lengths[320];  // code lengths of the block being set up: literals / lengths, then distances
	uint16_t codes[320];   // their canonical codes, bits reversed (the stream delivers a code with its first bit lowest)
	uint32_t event[8];     // what the decoding lane tells the others: kind, position, length, distance, error
	// the small tables of the decoding lane while a code is set up (in registers they are indexed dynamically and cost the kernel 60 VGPRs, i.e. a wavefront per SIMD)
	uint16_t work_count[16], work_offset[16], work_symbols[19]; uint32_t work_next_code[16]; uint8_t work_code_lengths[19];
};

// the bits of the stream, lowest first.  The input is read in 4-byte words, and always one word ahead of the one that is needed: the load of word k + 1 is issued when word k goes
// into the buffer, so that its latency (a chain of dependent loads from HBM would cost ~1 us per word: 30 ms for a block of 16 KB, as the first version of this reader did byte by
// byte) lies under the decoding of the 32 bits before it.  Reads up to 12 bytes behind the end of the block: the caller's buffer is padded.
struct InflateBits {
	const uint8_t* bytes; uint32_t size, at; // `at`: bytes consumed into `buffer`
	unsigned long long buffer; uint32_t count;
	uint32_t ahead; // the word at `at`, loaded already
	AGPU_HD static uint32_t word_at(const uint8_t* source) { uint32_t word; __builtin_memcpy(&word, source, 4); return word; }
	AGPU_HD void start(const uint8_t* source, uint32_t n) { bytes = source; size = n; at = 0; buffer = 0; count = 0; ahead = word_at(bytes); }
	AGPU_HD void refill() { // at least 32 bits afterwards (what lies behind the end of the input is never decoded into output: a code that runs over it is caught by `overrun`)
		if (count <= 32) { buffer |= (unsigned long long) ahead << count; count += 32; at += 4; ahead = word_at(bytes + at); }
	}
	AGPU_HD uint32_t peek(int n) const { return (uint32_t) (buffer & ((1ull << n) - 1ull)); }
	AGPU_HD void drop(int n) { buffer >>= n; count -= n; }
	AGPU_HD uint32_t take(int n) { const uint32_t value = peek(n); drop(n); return value; }
	AGPU_HD void to_byte_boundary() { drop((int) (count & 7u)); }
	AGPU_HD bool overrun() const { return at > size && (at - size) * 8 > count; } // more bits were taken than the input holds
};

AGPU_HD uint32_t inflate_reverse_bits(uint32_t code, int length) { uint32_t reversed = 0; for (int k = 0; k < length; ++k) { reversed = reversed << 1 | (code & 1u); code >>= 1; } return reversed; }

// one symbol: the first-level table, else bit by bit over the canonical counts; returns -1 for a code that does not exist
AGPU_HD int inflate_symbol(InflateBits& bits, const uint16_t* fast, int fast_bits, const InflateHuffman& huffman) {
	const uint32_t entry = fast[bits.peek(fast_bits)];
	if (entry != 0) { bits.drop((int) (entry & 15u)); return (int) (entry >> 4); }
	int code = 0, first = 0, index = 0;
	for (int length = 1; length <= 15; ++length) {
		code |= (int) bits.take(1);
		const int count = huffman.count[length];
		if (code - count < first) return huffman.symbol[index + (code - first)];
		index += count; first += count; first <<= 1; code <<= 1;
	}
	return -1;
}

// the tables of a code from the lengths of its symbols (lengths[0..n)); lane 0 numbers the codes, all lanes fill the first-level table.  Returns false for an
// over-subscribed or incomplete set of lengths (an incomplete one is allowed where zlib allows it: a single code of one bit, or none)
template <class Sync> AGPU_HD bool inflate_build(const uint8_t* lengths, uint32_t n, uint16_t* codes, InflateHuffman& huffman, uint16_t* fast, int fast_bits, uint32_t lane, uint32_t lanes, uint32_t* verdict, uint16_t* offset, uint32_t* next_code, Sync sync) {
	for (uint32_t k = lane; k < (1u << fast_bits); k += lanes) fast[k] = 0;
	if (lane == 0) {
		for (int length = 0; length <= 15; ++length) huffman.count[length] = 0;
		for (uint32_t s = 0; s < n; ++s) huffman.count[lengths[s]]++;
		int left = 1, longest = 0; bool valid = true;
		for (int length = 1; length <= 15; ++length) { if (left >= 0) { left <<= 1; left -= huffman.count[length]; } if (left < 0) valid = false; if (huffman.count[length] != 0) longest = length; }
		if (left > 0 && longest > 1) valid = false; // (advisor, round 4: zlib refuses an incomplete set too, unless it is a single code of one bit -- or no code at all; inftrees.c)
		offset[1] = 0;
		for (int length = 1; length < 15; ++length) offset[length + 1] = offset[length] + huffman.count[length];
		uint32_t code = 0;
		for (int length = 1; length <= 15; ++length) { code = (code + (length > 1 ? huffman.count[length - 1] : 0)) << 1; next_code[length] = code; }
		for (uint32_t s = 0; s < n; ++s) {
			const int length = lengths[s];
			if (length == 0) continue;
			huffman.symbol[offset[length]++] = (uint16_t) s;
			codes[s] = (uint16_t) inflate_reverse_bits(next_code[length]++, length);
		}
		huffman.count[0] = 0;
		*verdict = valid ? 1u : 0u;
	}
	sync();
	for (uint32_t s = lane; s < n; s += lanes) {
		const uint32_t length = lengths[s];
		if (length == 0 || length > (uint32_t) fast_bits) continue;
		for (uint32_t k = codes[s]; k < (1u << fast_bits); k += 1u << length) fast[k] = (uint16_t) (s << 4 | length);
	}
	sync();
	return *verdict != 0;
}

// The block.  `output` receives out_size bytes (the ISIZE of the gzip trailer); returns INFLATE_OK or what was wrong with the stream.  `sync` orders the lanes (a
// barrier of the workgroup on the device, nothing on the host); `broadcast(value)` gives every lane lane 0's value.
template <class Sync, class Broadcast> AGPU_HD int inflate_block(const uint8_t* input, uint32_t in_size, uint8_t* output, uint32_t out_size, InflateShared& shared, uint32_t lane, uint32_t lanes, Sync sync, Broadcast broadcast) {
	static const uint16_t length_base[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
	static const uint8_t length_extra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
	static const uint16_t distance_base[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
	static const uint8_t distance_extra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
	static const uint8_t length_order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
	enum { EVENT_NONE = 0, EVENT_MATCH = 1, EVENT_FLUSH = 2, EVENT_TABLES = 3, EVENT_END = 4, EVENT_ERROR = 5 };
	InflateBits bits; bits.start(input, in_size); // (only lane 0's copy moves)
	uint32_t produced = 0, flushed = 0; // bytes decoded / written back to `output`, the same on every lane between the events
	bool last_block = false, in_block = false; uint32_t stored_left = 0; bool stored = false; // lane 0's view of the DEFLATE block it is in
	auto write_back = [&](uint32_t until) { // all lanes: ring -> output for [flushed, until)
		for (uint32_t k = flushed + lane; k < until; k += lanes) output[k] = shared.ring[k & (INFLATE_RING - 1)];
		flushed = until;
	};
	while (true) {
		if (lane == 0) { // decode until something needs all lanes
			uint32_t kind = EVENT_NONE, length = 0, distance = 0, error = INFLATE_OK;
			while (kind == EVENT_NONE) {
				bits.refill();
				if (bits.overrun()) { kind = EVENT_ERROR; error = INFLATE_INPUT_OVERRUN; break; }
				if (!in_block) {
					if (last_block) { kind = EVENT_END; break; }
					last_block = bits.take(1) != 0;
					const uint32_t type = bits.take(2);
					if (type == 0) { // stored: LEN, NLEN, then LEN bytes as they are
						bits.to_byte_boundary(); bits.refill();
						const uint32_t len = bits.take(16), nlen = bits.take(16);
						if ((len ^ 0xFFFFu) != nlen) { kind = EVENT_ERROR; error = INFLATE_BAD_STORED_LENGTH; break; }
						stored = true; stored_left = len; in_block = true;
					} else if (type == 1) { // fixed code (RFC 1951 3.2.6)
						for (uint32_t s = 0; s < 288; ++s) shared.lengths[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
						for (uint32_t s = 0; s < 32; ++s) shared.lengths[288 + s] = 5; // (32 codes of 5 bits, the last two never used: a complete code, as zlib's fixed table)
						shared.event[5] = 288; shared.event[6] = 32;
						stored = false; in_block = true; kind = EVENT_TABLES;
					} else if (type == 2) { // dynamic code: the lengths of the code lengths' code, then the lengths of both codes, run-length coded (3.2.7)
						const uint32_t n_litlen = bits.take(5) + 257, n_distance = bits.take(5) + 1, n_lengths = bits.take(4) + 4;
						if (n_litlen > 286 || n_distance > 30) { kind = EVENT_ERROR; error = INFLATE_BAD_CODE_LENGTHS; break; }
						uint8_t* code_lengths = shared.work_code_lengths;
						for (uint32_t k = 0; k < 19; ++k) code_lengths[k] = 0;
						for (uint32_t k = 0; k < n_lengths; ++k) { bits.refill(); code_lengths[length_order[k]] = (uint8_t) bits.take(3); }
						// (a code of 19 symbols: decoded over its canonical counts, no table)
						uint16_t* count = shared.work_count; uint16_t* offset = shared.work_offset; uint16_t* symbols = shared.work_symbols;
						for (int l = 0; l <= 15; ++l) count[l] = 0;
						for (uint32_t k = 0; k < 19; ++k) count[code_lengths[k]]++;
						count[0] = 0;
						int left = 1; bool valid = true;
						for (int l = 1; l <= 15; ++l) { if (left >= 0) { left <<= 1; left -= count[l]; } if (left < 0) valid = false; }
						if (!valid || left > 0) { kind = EVENT_ERROR; error = INFLATE_BAD_CODE_LENGTHS; break; } // (zlib: the code of the code lengths must be complete)
						offset[1] = 0;
						for (int l = 1; l < 15; ++l) offset[l + 1] = offset[l] + count[l];
						for (uint32_t k = 0; k < 19; ++k) if (code_lengths[k] != 0) symbols[offset[code_lengths[k]]++] = (uint16_t) k;
						uint32_t filled = 0;
						while (filled < n_litlen + n_distance && error == INFLATE_OK) {
							bits.refill();
							int code = 0, first = 0, index = 0, symbol = -1;
							for (int l = 1; l <= 15; ++l) {
								code |= (int) bits.take(1);
								if (code - (int) count[l] < first) { symbol = symbols[index + (code - first)]; break; }
								index += count[l]; first += count[l]; first <<= 1; code <<= 1;
							}
							if (symbol < 0) { error = INFLATE_BAD_CODE_LENGTHS; break; }
							if (symbol < 16) { shared.lengths[filled++] = (uint8_t) symbol; continue; }
							uint32_t repeat, value = 0;
							if (symbol == 16) { if (filled == 0) { error = INFLATE_BAD_CODE_LENGTHS; break; } value = shared.lengths[filled - 1]; repeat = 3 + bits.take(2); }
							else if (symbol == 17) repeat = 3 + bits.take(3);
							else repeat = 11 + bits.take(7);
							if (filled + repeat > n_litlen + n_distance) { error = INFLATE_BAD_CODE_LENGTHS; break; }
							while (repeat-- > 0) shared.lengths[filled++] = (uint8_t) value;
						}
						if (error == INFLATE_OK && shared.lengths[256] == 0) error = INFLATE_BAD_CODE_LENGTHS; // no end-of-block code
						if (error != INFLATE_OK) { kind = EVENT_ERROR; break; }
						// (the distance lengths behind the literal / length ones: moved to their place at 288)
						for (uint32_t k = n_distance; k-- > 0; ) shared.lengths[288 + k] = shared.lengths[n_litlen + k];
						for (uint32_t k = n_litlen; k < 288; ++k) shared.lengths[k] = 0;
						shared.event[5] = 288; shared.event[6] = n_distance;
						stored = false; in_block = true; kind = EVENT_TABLES;
					} else { kind = EVENT_ERROR; error = INFLATE_BAD_BLOCK_TYPE; }
					continue;
				}
				if (stored) {
					if (stored_left == 0) { in_block = false; continue; }
					if (produced >= out_size) { kind = EVENT_ERROR; error = INFLATE_OUTPUT_OVERRUN; break; }
					shared.ring[produced & (INFLATE_RING - 1)] = (uint8_t) bits.take(8); ++produced; --stored_left;
					if (produced - flushed >= INFLATE_FLUSH) kind = EVENT_FLUSH;
					continue;
				}
				const int symbol = inflate_symbol(bits, shared.litlen_fast, INFLATE_LITLEN_BITS, shared.litlen);
				if (symbol < 0) { kind = EVENT_ERROR; error = INFLATE_BAD_SYMBOL; break; }
				if (symbol < 256) {
					if (produced >= out_size) { kind = EVENT_ERROR; error = INFLATE_OUTPUT_OVERRUN; break; }
					shared.ring[produced & (INFLATE_RING - 1)] = (uint8_t) symbol; ++produced;
					if (produced - flushed >= INFLATE_FLUSH) kind = EVENT_FLUSH;
					continue;
				}
				if (symbol == 256) { in_block = false; continue; }
				if (symbol > 285) { kind = EVENT_ERROR; error = INFLATE_BAD_SYMBOL; break; }
				bits.refill();
				length = length_base[symbol - 257] + bits.take(length_extra[symbol - 257]);
				const int distance_symbol = inflate_symbol(bits, shared.distance_fast, INFLATE_DISTANCE_BITS, shared.distance);
				if (distance_symbol < 0 || distance_symbol > 29) { kind = EVENT_ERROR; error = INFLATE_BAD_DISTANCE; break; }
				bits.refill();
				distance = distance_base[distance_symbol] + bits.take(distance_extra[distance_symbol]);
				if (distance > produced) { kind = EVENT_ERROR; error = INFLATE_BAD_DISTANCE; break; }
				if (produced + length > out_size) { kind = EVENT_ERROR; error = INFLATE_OUTPUT_OVERRUN; break; }
				// a short match from inside the ring is copied by the decoding lane itself: most matches of a BAM block at the usual levels are a few bytes long, and handing every
				// one of them to all lanes (three barriers and a broadcast) cost more than the copy (profiles/r04e_bench100m.json: 17 ms per block)
				if (length <= INFLATE_SHORT_MATCH && distance + length <= INFLATE_RING / 2) {
					for (uint32_t i = 0; i < length; ++i) shared.ring[(produced + i) & (INFLATE_RING - 1)] = shared.ring[(produced + i - distance) & (INFLATE_RING - 1)];
					produced += length;
					if (produced - flushed >= INFLATE_FLUSH) kind = EVENT_FLUSH;
					continue;
				}
				kind = EVENT_MATCH;
			}
			shared.event[0] = kind; shared.event[1] = produced; shared.event[2] = length; shared.event[3] = distance; shared.event[4] = error;
		}
		sync();
		const uint32_t kind = broadcast(shared.event[0]);
		produced = broadcast(shared.event[1]);
		if (kind == EVENT_ERROR) return (int) broadcast(shared.event[4]);
		if (kind == EVENT_END) {
			write_back(produced);
			return produced == out_size ? INFLATE_OK : INFLATE_SIZE_MISMATCH;
		}
		if (kind == EVENT_TABLES) {
			uint32_t* verdict = &shared.event[7];
			const bool litlen_valid = inflate_build(shared.lengths, 288, shared.codes, shared.litlen, shared.litlen_fast, INFLATE_LITLEN_BITS, lane, lanes, verdict, shared.work_offset, shared.work_next_code, sync);
			const bool distance_valid = inflate_build(shared.lengths + 288, broadcast(shared.event[6]), shared.codes, shared.distance, shared.distance_fast, INFLATE_DISTANCE_BITS, lane, lanes, verdict, shared.work_offset, shared.work_next_code, sync);
			if (!litlen_valid || !distance_valid) return INFLATE_BAD_CODE_LENGTHS;
		} else if (kind == EVENT_MATCH) {
			const uint32_t length = broadcast(shared.event[2]), distance = broadcast(shared.event[3]), from = produced - distance;
			for (uint32_t i = lane; i < length; i += lanes) {
				const uint32_t source = from + (distance >= length ? i : i % distance);
				// in the ring as long as nothing has been written over it: the ring holds [produced + length - RING, produced + length) at most while this match is written
				const uint8_t byte = source + INFLATE_RING >= produced + length ? shared.ring[source & (INFLATE_RING - 1)] : output[source];
				shared.ring[(produced + i) & (INFLATE_RING - 1)] = byte;
			}
			produced += length;
			sync();
			if (produced - flushed >= INFLATE_FLUSH) { write_back(produced); sync(); }
			if (lane == 0) shared.event[1] = produced;
			// (lane 0 continues with its own `produced`, which it advances here as everybody does)
		} else if (kind == EVENT_FLUSH) {
			write_back(produced);
			sync();
		}
		sync();
	}
}

}

#endif
