// inflate_fast_core.hpp -- the DEFLATE stream of a BGZF block in two passes, for many blocks at a time (round 5; inflate_core.hpp is the one-wavefront-per-block decoder of
// round 4, kept as the second implementation and as the way out for the few blocks this one hands back).
//
// Why two passes.  Decoding a DEFLATE stream is two different kinds of work: the Huffman codes are a chain (where a code ends is known only when it has been looked up), the
// copies of the matches are not.  Round 4 gave a whole wavefront to one block: one lane walked the chain, 63 waited, and every match of more than a few bytes cost three barriers:
// 5.9 GB/s of output on the whole device.  Here
//   pass 1 (inflate_tokens)   ONE LANE per block, sixteen blocks per wavefront running the same loop on different data: bits from a 64-bit buffer refilled without a branch from a
//                             ring of input bytes in LDS (topped up from HBM once in eight tokens, by all lanes at once), literal / length codes and distance codes through two-level tables (8 and 7 bits, then as many more as the longest code of the prefix needs), base values and extra bits of lengths and distances computed, not
//                             looked up.  Literals go straight to their place in the output; a match is only NOTED: (position, length, distance), 8 bytes.
//   pass 2 (resolve)          a wavefront per block takes the noted matches 64 at a time, a lane per match.  A match may be copied as soon as the bytes it reads are final:
//                             everything in front of the first match that is still pending is (inflate_match_is_ready) -- most matches of a group read from far in front of the
//                             group and go in the first round.
// The tables of a block are 2 KB of LDS, so 80 blocks are decoded per CU at a time instead of 12, and the lanes of a wavefront no longer wait for one another's barriers.
//
// The bytes are those zlib's inflate gives, and the streams zlib refuses are refused (over-subscribed AND incomplete codes: inftrees.c's rule, which the decoder of round 4 did not
// have); tests/emu/inflate_check.cpp steps both passes on the host against zlib on every kind of block.  A block with more matches than INFLATE_MATCH_CAPACITY returns INFLATE_RETRY
// and goes through the decoder of round 4.
#ifndef AGPU_INFLATE_FAST_CORE_HPP
#define AGPU_INFLATE_FAST_CORE_HPP 1

#include "inflate_core.hpp"

namespace agpu {

// Two-level tables for both codes: `root` bits at once, the longer codes through a second table per prefix (as wide as the longest code under that prefix needs).  The sizes are
// what a decoding lane can afford in LDS -- 80 lanes per CU at 2 032 bytes each -- not what the worst code needs: a block whose second tables do not fit (zlib's "enough" says
// 852 entries for 9 bits; the blocks of a BAM file at the usual levels need ~250 of the 384 behind 8 bits, and ~20 of the 128 behind the 7 bits of the distances) is handed back.
const uint32_t FAST_LITLEN_ROOT = 8, FAST_LITLEN_ENTRIES = 256 + 384, FAST_DISTANCE_ROOT = 7, FAST_DISTANCE_ENTRIES = 128 + 128;
const uint32_t INFLATE_MATCH_CAPACITY = 8192; // matches noted per block (64 KB of notes for 64 KB of output); a 64 KB BAM block at the usual levels has 4 000 - 6 000
enum { INFLATE_RETRY = 9 };

// table entries: payload << 6 | type << 4 | bits.  0 = no such code.
enum { FAST_LITERAL = 1 /* payload = the byte; in the table of the distances: the distance symbol */, FAST_LENGTH = 2 /* payload = symbol - 256: 0 end of block, 1..29 a length code */, FAST_SUBTABLE = 3 /* payload = where it starts, bits = how many bits index it */ };
AGPU_HD uint16_t fast_entry(uint32_t type, uint32_t payload, uint32_t n_bits) { return (uint16_t) (payload << 6 | type << 4 | n_bits); }

struct alignas(16) InflateFastTables { // what one decoding lane keeps in LDS: 2 032 bytes
	unsigned long long input_ring[18];          // the next 128 bytes of the block's DEFLATE stream (FastBits) + two spare words
	uint16_t litlen[FAST_LITLEN_ENTRIES];       // (while a dynamic header is read: the 7-bit table of the code-length code, as bytes)
	uint16_t distance[FAST_DISTANCE_ENTRIES];   // (until it is built: the code lengths of the block being set up, as bytes -- literals / lengths, then distances)
	uint16_t count[16], next[16];               // codes per length, the next canonical code of every length to hand out
	uint8_t code_lengths[32];                   // lengths of the code-length code; then the lengths of the distance code while its table is built
};

// the bits of the stream, lowest first; at least 56 of them after refill().  The input comes through a ring of 128 bytes in LDS that is topped up from HBM in pieces of 16 bytes
// -- not at every token: the first version loaded the 8 bytes of the next token from HBM a token ahead, and the `s_waitcnt vmcnt(0)` in front of their use also waited for the
// store of the token before (a literal, a noted match): a round trip to the L2 per token, 0.9 us of it (profiles/r05b: 20.7 ms per piece of 17 000 blocks).  vmcnt counts loads
// and stores alike, so the loop of the tokens must not load from HBM at all: refill() reads the ring, the stores are never waited for, and top_up() -- every eighth token, at the
// same moment in all lanes of the wavefront -- is the only place that waits.  Reads up to 208 bytes behind the end of the input (the callers pad by 256).
const uint32_t INFLATE_INPUT_RING = 128;
struct FastBits {
	const uint8_t* input; uint32_t size, next /* bytes taken into the buffer */, loaded /* the ring holds the bytes [loaded - 128, loaded) of the input */;
	unsigned long long* ring;
	unsigned long long buffer; uint32_t count;
	AGPU_HD static unsigned long long load64(const uint8_t* p) { unsigned long long word; __builtin_memcpy(&word, p, 8); return word; }
	AGPU_HD void top_up() { // afterwards more than 112 bytes lie ahead of `next`: 8 tokens of at most 48 bits, or 64 symbols of a header, before the next call
		while (loaded - next <= INFLATE_INPUT_RING - 16) {
			// four pieces are loaded whether they are wanted or not (one wait for all of them; what lies behind the input is padding), those that fit go into the ring
			struct Pair { unsigned long long low, high; } piece0, piece1, piece2, piece3;
			__builtin_memcpy(&piece0, input + loaded, 16); __builtin_memcpy(&piece1, input + loaded + 16, 16);
			__builtin_memcpy(&piece2, input + loaded + 32, 16); __builtin_memcpy(&piece3, input + loaded + 48, 16);
			const uint32_t fit = (INFLATE_INPUT_RING - (loaded - next)) >> 4; // >= 1
			const uint32_t word = loaded >> 3, mask = INFLATE_INPUT_RING / 8 - 1; // (16 bytes at a multiple of 16: never across the end of the ring)
			// (a piece that does not fit is written to the two spare words behind the ring: every load is used inside this branch, so the compiler waits for it here and not -- for
			//  the registers it would land in -- at the place where the branch joins the loop of the tokens, token after token)
			const uint32_t at1 = fit > 1 ? (word + 2) & mask : INFLATE_INPUT_RING / 8, at2 = fit > 2 ? (word + 4) & mask : INFLATE_INPUT_RING / 8, at3 = fit > 3 ? (word + 6) & mask : INFLATE_INPUT_RING / 8;
			ring[word & mask] = piece0.low; ring[(word & mask) + 1] = piece0.high;
			ring[at1] = piece1.low; ring[at1 + 1] = piece1.high;
			ring[at2] = piece2.low; ring[at2 + 1] = piece2.high;
			ring[at3] = piece3.low; ring[at3 + 1] = piece3.high;
			loaded += 16 * (fit < 4 ? fit : 4);
		}
	}
	AGPU_HD void start(const uint8_t* bytes, uint32_t n, unsigned long long* words) { input = bytes; size = n; next = 0; loaded = 0; ring = words; buffer = 0; count = 0; top_up(); }
	AGPU_HD void refill() { // (the bits above `count` that the shift leaves in the buffer are the low bits of the byte at `next`: the next refill puts the same bits there again)
		const uint32_t word = next >> 3, shift = (next & 7u) << 3;
		const unsigned long long low = ring[word & (INFLATE_INPUT_RING / 8 - 1)], high = ring[(word + 1) & (INFLATE_INPUT_RING / 8 - 1)];
		buffer |= ((low >> shift) | ((high << 1) << (63u - shift))) << count;
		const uint32_t bytes = (63u - count) >> 3;
		next += bytes; count += bytes << 3;
	}
	AGPU_HD void refill_anywhere() { if (loaded - next < 32) top_up(); refill(); } // (outside the loop of the tokens: headers, stored blocks)
	AGPU_HD uint32_t peek(uint32_t n) const { return (uint32_t) buffer & ((1u << n) - 1u); } // n <= 16
	AGPU_HD void drop(uint32_t n) { buffer >>= n; count -= n; }
	AGPU_HD uint32_t take(uint32_t n) { const uint32_t value = peek(n); drop(n); return value; }
	AGPU_HD bool overrun() const { return next > size && (next - size) * 8u > count; } // more bits were taken than the input holds
};

AGPU_HD uint32_t fast_reverse_bits(uint32_t code, uint32_t length) { // the stream delivers a code with its first bit lowest
#if defined(__clang__)
	return __builtin_bitreverse32(code) >> (32u - length);
#else
	uint32_t reversed = 0;
	for (uint32_t k = 0; k < length; ++k) { reversed = reversed << 1 | (code & 1u); code >>= 1; }
	return reversed;
#endif
}

// counts per length and the check zlib makes (inftrees.c: over-subscribed sets are refused, incomplete ones too unless the code has a single symbol of one bit -- or none at all).
// Returns the longest length, -1 for a set zlib refuses.
AGPU_HD int fast_count_lengths(const uint8_t* lengths, uint32_t n, InflateFastTables& t) {
	for (uint32_t l = 0; l < 16; ++l) t.count[l] = 0;
	for (uint32_t s = 0; s < n; ++s) t.count[lengths[s]]++;
	t.count[0] = 0;
	int left = 1, longest = 0;
	for (uint32_t l = 1; l < 16; ++l) {
		left = (left << 1) - (int) t.count[l];
		if (left < 0) return -1;
		if (t.count[l] != 0) longest = (int) l;
	}
	if (left > 0 && longest > 1) return -1;
	return longest;
}
AGPU_HD void fast_first_codes(InflateFastTables& t) { // the first canonical code of every length (RFC 1951 3.2.2)
	uint32_t code = 0;
	t.next[0] = 0;
	for (uint32_t l = 1; l < 16; ++l) { code = (code + t.count[l - 1]) << 1; t.next[l] = (uint16_t) code; }
}
// the table of one code.  litlen: the symbols below 256 are literals, the others lengths / the end of the block; else every symbol is a distance symbol.
AGPU_HD int fast_build_table(const uint8_t* lengths, uint32_t n, uint16_t* table, uint32_t root, uint32_t capacity, bool litlen, InflateFastTables& t) {
	const int longest = fast_count_lengths(lengths, n, t);
	if (longest < 0) return INFLATE_BAD_CODE_LENGTHS;
	uint32_t* words = (uint32_t*) table;
	for (uint32_t k = 0; k < (1u << root) / 2; ++k) words[k] = 0;
	if (longest > (int) root) { // how wide the second table of every prefix is
		fast_first_codes(t);
		for (uint32_t s = 0; s < n; ++s) {
			const uint32_t length = lengths[s];
			if (length == 0) continue;
			const uint32_t code = t.next[length]++;
			if (length <= root) continue;
			const uint32_t prefix = fast_reverse_bits(code, length) & ((1u << root) - 1u);
			if ((table[prefix] & 15u) < length - root) table[prefix] = fast_entry(FAST_SUBTABLE, 0, length - root);
		}
	}
	fast_first_codes(t);
	uint32_t next_free = 1u << root;
	for (uint32_t s = 0; s < n; ++s) {
		const uint32_t length = lengths[s];
		if (length == 0) continue;
		const uint32_t reversed = fast_reverse_bits(t.next[length]++, length);
		const uint32_t type = !litlen || s < 256 ? FAST_LITERAL : FAST_LENGTH, payload = !litlen || s < 256 ? s : s - 256;
		if (length <= root) {
			const uint16_t entry = fast_entry(type, payload, length);
			for (uint32_t k = reversed; k < (1u << root); k += 1u << length) table[k] = entry;
			continue;
		}
		const uint32_t prefix = reversed & ((1u << root) - 1u);
		const uint32_t bits = table[prefix] & 15u;
		uint32_t offset = table[prefix] >> 6;
		if (offset == 0) {
			offset = next_free; next_free += 1u << bits;
			if (next_free > capacity) return INFLATE_RETRY; // (more second tables than a lane has room for: the other decoder takes the block)
			table[prefix] = fast_entry(FAST_SUBTABLE, offset, bits);
		}
		const uint16_t entry = fast_entry(type, payload, length - root);
		for (uint32_t k = reversed >> root; k < (1u << bits); k += 1u << (length - root)) table[offset + k] = entry;
	}
	return INFLATE_OK;
}
// one symbol: the entry of its code (0: no such code), its bits dropped
AGPU_HD uint32_t fast_symbol(FastBits& bits, const uint16_t* table, uint32_t root) {
	uint32_t entry = table[bits.peek(root)];
	if ((entry >> 4 & 3u) == FAST_SUBTABLE) { bits.drop(root); entry = table[(entry >> 6) + bits.peek(entry & 15u)]; }
	bits.drop(entry & 15u);
	return entry;
}

// the header of a dynamic block (RFC 1951 3.2.7): the lengths of both codes, run-length coded with a code of 19 symbols, itself given by 3-bit lengths
AGPU_HD int fast_read_dynamic_header(FastBits& bits, InflateFastTables& t, uint32_t& n_litlen, uint32_t& n_distance) {
	static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
	bits.refill_anywhere();
	n_litlen = bits.take(5) + 257; n_distance = bits.take(5) + 1;
	const uint32_t n_code_lengths = bits.take(4) + 4;
	if (n_litlen > 286 || n_distance > 30) return INFLATE_BAD_CODE_LENGTHS;
	for (uint32_t k = 0; k < 19; ++k) t.code_lengths[k] = 0;
	for (uint32_t k = 0; k < n_code_lengths; ++k) { if ((k & 7u) == 0) bits.refill_anywhere(); t.code_lengths[order[k]] = (uint8_t) bits.take(3); }
	if (bits.overrun()) return INFLATE_INPUT_OVERRUN;
	// the code of the code lengths must be complete (zlib); 7 bits at once
	for (uint32_t l = 0; l < 8; ++l) t.count[l] = 0;
	for (uint32_t k = 0; k < 19; ++k) t.count[t.code_lengths[k]]++;
	t.count[0] = 0;
	int left = 1;
	for (uint32_t l = 1; l < 8; ++l) { left = (left << 1) - (int) t.count[l]; if (left < 0) return INFLATE_BAD_CODE_LENGTHS; }
	if (left > 0) return INFLATE_BAD_CODE_LENGTHS;
	uint32_t code = 0;
	t.next[0] = 0;
	for (uint32_t l = 1; l < 8; ++l) { code = (code + t.count[l - 1]) << 1; t.next[l] = (uint16_t) code; }
	uint8_t* table = (uint8_t*) t.litlen; uint8_t* lengths = (uint8_t*) t.distance;
	for (uint32_t k = 0; k < 19; ++k) {
		const uint32_t length = t.code_lengths[k];
		if (length == 0) continue;
		for (uint32_t j = fast_reverse_bits(t.next[length]++, length); j < 128; j += 1u << length) table[j] = (uint8_t) (k << 3 | length);
	}
	const uint32_t total = n_litlen + n_distance;
	uint32_t filled = 0;
	while (filled < total) {
		bits.refill_anywhere();
		if (bits.overrun()) return INFLATE_INPUT_OVERRUN; // (advisor, round 4: the loops of the header checked nothing)
		const uint32_t entry = table[bits.peek(7)];
		bits.drop(entry & 7u);
		const uint32_t symbol = entry >> 3;
		if (symbol < 16) { lengths[filled++] = (uint8_t) symbol; continue; }
		uint32_t repeat, value = 0;
		if (symbol == 16) { if (filled == 0) return INFLATE_BAD_CODE_LENGTHS; value = lengths[filled - 1]; repeat = 3 + bits.take(2); }
		else if (symbol == 17) repeat = 3 + bits.take(3);
		else repeat = 11 + bits.take(7);
		if (filled + repeat > total) return INFLATE_BAD_CODE_LENGTHS;
		while (repeat-- > 0) lengths[filled++] = (uint8_t) value;
	}
	if (lengths[256] == 0) return INFLATE_BAD_CODE_LENGTHS; // no end-of-block code
	return INFLATE_OK;
}

// a noted match: position of its first byte in the block's output | length << 16 | distance << 32
AGPU_HD unsigned long long inflate_match_note(uint32_t position, uint32_t length, uint32_t distance) { return (unsigned long long) position | (unsigned long long) length << 16 | (unsigned long long) distance << 32; }
AGPU_HD uint32_t inflate_note_position(unsigned long long note) { return (uint32_t) note & 0xFFFFu; }
AGPU_HD uint32_t inflate_note_length(unsigned long long note) { return (uint32_t) (note >> 16) & 0xFFFFu; }
AGPU_HD uint32_t inflate_note_distance(unsigned long long note) { return (uint32_t) (note >> 32); }

// Pass 1, one lane: the literals of the block into `output` (out_size bytes: the ISIZE of the gzip trailer), its matches into `notes`.
AGPU_HD int inflate_tokens(const uint8_t* input, uint32_t in_size, uint8_t* output, uint32_t out_size, unsigned long long* notes, uint32_t capacity, uint32_t& n_notes, InflateFastTables& t) {
	FastBits bits; bits.start(input, in_size, t.input_ring);
	uint32_t produced = 0, noted = 0;
	n_notes = 0;
	bool last_block = false;
	while (!last_block) {
		bits.refill_anywhere();
		if (bits.overrun()) return INFLATE_INPUT_OVERRUN;
		last_block = bits.take(1) != 0;
		const uint32_t type = bits.take(2);
		if (type == 0) { // stored: LEN, NLEN, then LEN bytes as they are
			bits.drop(bits.count & 7u); bits.refill_anywhere();
			const uint32_t length = bits.take(16), complement = bits.take(16);
			if ((length ^ 0xFFFFu) != complement) return INFLATE_BAD_STORED_LENGTH;
			if (produced + length > out_size) return INFLATE_OUTPUT_OVERRUN;
			for (uint32_t k = 0; k < length; ++k) {
				if ((k & 3u) == 0) { bits.refill_anywhere(); if (bits.overrun()) return INFLATE_INPUT_OVERRUN; }
				output[produced++] = (uint8_t) bits.take(8);
			}
			continue;
		}
		if (type == 3) return INFLATE_BAD_BLOCK_TYPE;
		uint32_t n_litlen = 288, n_distance = 32; // fixed code (3.2.6): 288 and 32 symbols (the last two of each never occur in a valid stream: refused when they are decoded)
		uint8_t* lengths = (uint8_t*) t.distance;
		if (type == 1) {
			for (uint32_t s = 0; s < 288; ++s) lengths[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
			for (uint32_t s = 0; s < 32; ++s) lengths[288 + s] = 5;
		} else { const int status = fast_read_dynamic_header(bits, t, n_litlen, n_distance); if (status != INFLATE_OK) return status; }
		{ const int status = fast_build_table(lengths, n_litlen, t.litlen, FAST_LITLEN_ROOT, FAST_LITLEN_ENTRIES, true, t); if (status != INFLATE_OK) return status; }
		for (uint32_t s = 0; s < n_distance; ++s) t.code_lengths[s] = lengths[n_litlen + s]; // (out of the way of the table that is built where they stand)
		{ const int status = fast_build_table(t.code_lengths, n_distance, t.distance, FAST_DISTANCE_ROOT, FAST_DISTANCE_ENTRIES, false, t); if (status != INFLATE_OK) return status; }
		// A token per turn: at most 15 + 5 + 15 + 13 = 48 bits.  What is wrong with a token is NOTED (`wrong`), not acted upon at once: a lane that leaves the loop in the middle of a turn
		// costs every turn of every lane the bookkeeping of who is still in it.  The note is looked at every eighth turn and at the end of the block; until then nothing is written
		// that should not be (a literal or a match that does not fit is not stored), and a code that does not exist is met again and again (its entry drops no bits).
		int wrong = INFLATE_OK;
		for (uint32_t turn = 0; ; ++turn) {
			if ((turn & 7u) == 0) { // (the lanes of a wavefront are in the same turn: they wait for HBM together, once in eight tokens)
				if (wrong != INFLATE_OK) return wrong;
				if (bits.overrun()) return INFLATE_INPUT_OVERRUN; // (at most eight tokens are decoded from what lies behind the input)
				bits.top_up();
			}
			bits.refill();
			const uint32_t entry = fast_symbol(bits, t.litlen, FAST_LITLEN_ROOT);
			const uint32_t payload = entry >> 6, type = entry >> 4 & 3u;
			if (type == FAST_LITERAL) {
				if (produced < out_size) output[produced++] = (uint8_t) payload; else wrong = INFLATE_OUTPUT_OVERRUN;
				continue;
			}
			if (type == FAST_LENGTH && payload == 0) break; // end of block
			const bool known = type == FAST_LENGTH && payload <= 29;
			const uint32_t s = known ? payload - 1 : 0; // 0..28 (3.2.5: lengths 3..258 in 29 codes, 0..5 extra bits)
			uint32_t length_extra = s < 8 ? 0 : (s >> 2) - 1, length = s < 4 ? 3 + s : 3 + ((4 + (s & 3u)) << length_extra);
			if (s == 28) { length = 258; length_extra = 0; }
			length += bits.take(length_extra);
			const uint32_t found = fast_symbol(bits, t.distance, FAST_DISTANCE_ROOT);
			const bool distance_known = found != 0 && (found >> 6) <= 29;
			const uint32_t symbol = distance_known ? found >> 6 : 0;
			const uint32_t distance_extra = symbol < 4 ? 0 : (symbol >> 1) - 1;
			const uint32_t distance = (symbol < 4 ? symbol + 1 : 1 + ((2 + (symbol & 1u)) << distance_extra)) + bits.take(distance_extra);
			const int verdict = !known ? (int) INFLATE_BAD_SYMBOL : !distance_known || distance > produced ? (int) INFLATE_BAD_DISTANCE : produced + length > out_size ? (int) INFLATE_OUTPUT_OVERRUN : noted == capacity ? (int) INFLATE_RETRY : (int) INFLATE_OK;
			if (verdict == INFLATE_OK) { notes[noted++] = inflate_match_note(produced, length, distance); produced += length; }
			else if (wrong == INFLATE_OK) wrong = verdict;
		}
		if (wrong != INFLATE_OK) return wrong;
	}
	if (bits.overrun()) return INFLATE_INPUT_OVERRUN;
	n_notes = noted;
	return produced == out_size ? INFLATE_OK : INFLATE_SIZE_MISMATCH;
}

// Pass 2.  A match reads the `min(length, distance)` bytes at its distance (a match that overlaps itself repeats them) and may be copied when they are final; `frontier` is the
// position of the first match (in the order of the stream) that has not been copied: every byte in front of it is a literal or belongs to a match that has been.
AGPU_HD bool inflate_match_is_ready(uint32_t position, uint32_t length, uint32_t distance, uint32_t frontier) {
	return position - distance + (length < distance ? length : distance) <= frontier;
}
// one match, by one lane: generations of `distance` bytes, all read from the same place (what stands at the distance when the match starts)
AGPU_HD void inflate_copy_match(uint8_t* output, uint32_t position, uint32_t length, uint32_t distance) {
	const uint8_t* source = output + position - distance;
	uint8_t* target = output + position;
	if (distance >= 8) {
		for (uint32_t done = 0; done < length; done += distance) {
			const uint32_t share = length - done < distance ? length - done : distance;
			uint32_t k = 0;
			for (; k + 8 <= share; k += 8) { const unsigned long long word = FastBits::load64(source + k); __builtin_memcpy(target + done + k, &word, 8); }
			if (k < share) { // (the 8 bytes read here may end up to 7 bytes behind the match's first byte: what lies there is not used, and the buffer of the output is padded)
				unsigned long long word = FastBits::load64(source + k);
				for (; k < share; ++k) { target[done + k] = (uint8_t) word; word >>= 8; }
			}
		}
	} else { // a period of fewer than 8 bytes: kept in a register
		unsigned long long pattern = 0;
		for (uint32_t k = 0; k < distance; ++k) pattern |= (unsigned long long) source[k] << (8 * k);
		uint32_t phase = 0;
		for (uint32_t k = 0; k < length; ++k) { target[k] = (uint8_t) (pattern >> (8 * phase)); if (++phase == distance) phase = 0; }
	}
}
// all matches in the order of the stream (the stepping harness; the device takes them 64 at a time, agpu_ingest.hip: bgzf_resolve_kernel)
AGPU_HD void inflate_resolve_in_order(uint8_t* output, const unsigned long long* notes, uint32_t n_notes) {
	for (uint32_t k = 0; k < n_notes; ++k) inflate_copy_match(output, inflate_note_position(notes[k]), inflate_note_length(notes[k]), inflate_note_distance(notes[k]));
}

}

#endif
