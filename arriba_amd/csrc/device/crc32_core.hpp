// crc32_core.hpp -- CRC-32 of the payload of a BGZF block (the check htslib makes on every block it reads: bgzf.c, check_header / inflate_block), for the blocks that go
// to the device as they are (stored blocks: agpu_ingest_push_bgzf).  A block is cut into chunks, every lane takes the CRC of one chunk with the byte-wise table, and the
// CRCs are joined pairwise: crc(A || B) = crc(A) advanced over |B| zero bytes, xor crc(B) -- the advance is a multiplication by x^(8|B|) modulo the polynomial, done with
// 32 x 32 bit matrices over GF(2) that are squared once per bit of |B| (the construction of zlib's crc32_combine).
#ifndef AGPU_CRC32_CORE_HPP
#define AGPU_CRC32_CORE_HPP 1

#include "views.hpp"

namespace agpu {

const uint32_t CRC32_POLYNOMIAL = 0xEDB88320u; // reflected

AGPU_HD uint32_t crc32_table_entry(uint32_t i) {
	uint32_t c = i;
	for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC32_POLYNOMIAL : c >> 1;
	return c;
}
// crc32(0, bytes, n) of zlib for one chunk: table[256] as crc32_table_entry gives it
AGPU_HD uint32_t crc32_of(const uint32_t* table, const uint8_t* bytes, size_t n) {
	uint32_t c = 0xFFFFFFFFu;
	for (size_t i = 0; i < n; ++i) c = table[(c ^ bytes[i]) & 0xFFu] ^ (c >> 8);
	return c ^ 0xFFFFFFFFu;
}
AGPU_HD uint32_t gf2_matrix_times(const uint32_t* matrix, uint32_t vector) {
	uint32_t sum = 0;
	for (int k = 0; vector != 0; vector >>= 1, ++k) if (vector & 1u) sum ^= matrix[k];
	return sum;
}
AGPU_HD void gf2_matrix_square(uint32_t* square, const uint32_t* matrix) { for (int k = 0; k < 32; ++k) square[k] = gf2_matrix_times(matrix, matrix[k]); }
// the same in parts: c = 0xFFFFFFFF before the first part, the CRC is ~c behind the last; every part but the last a multiple of four bytes
AGPU_HD uint32_t crc32_update_sliced(const uint32_t (*slice)[256], uint32_t c, const uint8_t* bytes, size_t n) {
	size_t i = 0;
	for (; i + 4 <= n; i += 4) {
		uint32_t word; __builtin_memcpy(&word, bytes + i, 4);
		c ^= word;
		c = slice[3][c & 0xFFu] ^ slice[2][(c >> 8) & 0xFFu] ^ slice[1][(c >> 16) & 0xFFu] ^ slice[0][c >> 24];
	}
	for (; i < n; ++i) c = slice[0][(c ^ bytes[i]) & 0xFFu] ^ (c >> 8);
	return c;
}
// crc32 of A || B from crc32(A), crc32(B) and |B| (zlib: crc32_combine)
AGPU_HD uint32_t crc32_joined(uint32_t crc_a, uint32_t crc_b, uint64_t length_b) {
	if (length_b == 0) return crc_a;
	uint32_t even[32], odd[32];
	odd[0] = CRC32_POLYNOMIAL; // the operator for one zero bit
	uint32_t row = 1;
	for (int k = 1; k < 32; ++k) { odd[k] = row; row <<= 1; }
	gf2_matrix_square(even, odd);  // two zero bits
	gf2_matrix_square(odd, even);  // four
	do { // the first square below gives the operator for one zero byte
		gf2_matrix_square(even, odd);
		if (length_b & 1u) crc_a = gf2_matrix_times(even, crc_a);
		length_b >>= 1;
		if (length_b == 0) break;
		gf2_matrix_square(odd, even);
		if (length_b & 1u) crc_a = gf2_matrix_times(odd, crc_a);
		length_b >>= 1;
	} while (length_b != 0);
	return crc_a ^ crc_b;
}

const uint32_t CRC32_CHUNK = 256; // bytes per lane and round of the device kernel

// What the device kernel works with (made once per context by crc32_make_tables): the tables of four bytes per step ("slicing by 4": slice[0] is the byte-wise table,
// slice[k][i] = the register k zero bytes further on), and for every power of two 2^k, k = 0 .. 16, the 32 x 32 matrix over GF(2) that advances a CRC register over 2^k
// zero bytes -- joining the CRCs of two pieces then costs one matrix-vector product per set bit of the length of the second piece, with the matrices in LDS, where
// crc32_joined squares them anew in the scratch memory of every lane (13 ms per 256 MB piece, four times the time the piece takes to arrive: profiles/r03h_*).
const int CRC32_ADVANCE_POWERS = 17; // pieces of up to 2^17 - 1 bytes (a BGZF block holds at most 2^16)
// Round 5: a block by ONE wavefront (bgzf_crc_kernel).  The payload is looked at as the END of a virtual block of 64 KB whose front is zero bytes; lane L takes the 1 KB chunk L
// of the virtual block, and all chunks are whole.  With the register started at 0 instead of 0xFFFFFFFF ("raw") zero bytes in front change nothing, and the standard start is the
// same as the first four bytes of the message inverted -- so crc32(M) = ~raw(zeros || M with its first four bytes inverted).  Raw CRCs join as crc32_joined does, and because
// every chunk is whole the distances are constants: the operator "over 1 KB of zero bytes" (and over 8 KB) as FOUR TABLES OF 256 WORDS, one look-up per byte of the register
// (join[.][b][v] = the operator applied to v << 8 b), instead of a 32-step matrix product per set bit of a length.  The 64 chunk CRCs are folded eight at a time (Horner: seven
// steps with the first operator, seven with the second).  Before: 256 lanes per block, chunks of 256 bytes, a tree of eight levels of matrix products, each level behind a
// barrier -- the joins took four times as long as the bytes (1.05 ms per 256 MB piece; 150 ms of a 10^8-fragment sample).
const uint32_t CRC32_WAVE_CHUNK = 1024, CRC32_WAVE_LANES = 64, CRC32_VIRTUAL = CRC32_WAVE_CHUNK * CRC32_WAVE_LANES, CRC32_ROUND_WORDS = 16;
struct Crc32Tables { uint32_t slice[4][256]; uint32_t advance[CRC32_ADVANCE_POWERS][32]; uint32_t join[2][4][256]; };
AGPU_HD uint32_t crc32_apply(const uint32_t (*op)[256], uint32_t c) { return op[0][c & 0xFFu] ^ op[1][(c >> 8) & 0xFFu] ^ op[2][(c >> 16) & 0xFFu] ^ op[3][c >> 24]; }
// four more bytes into a raw register (the step of crc32_update_sliced)
AGPU_HD uint32_t crc32_raw_step(const uint32_t (*slice)[256], uint32_t c, uint32_t word) {
	c ^= word;
	return slice[3][c & 0xFFu] ^ slice[2][(c >> 8) & 0xFFu] ^ slice[1][(c >> 16) & 0xFFu] ^ slice[0][c >> 24];
}
// word w of the virtual block of a payload of 4 <= n <= CRC32_VIRTUAL bytes: zero bytes in front, the first four bytes of the payload inverted; nothing outside the payload is read
AGPU_HD uint32_t crc32_virtual_word(const uint8_t* payload, uint32_t n, uint32_t w) {
	const int32_t m = (int32_t) (4 * w) - (int32_t) (CRC32_VIRTUAL - n); // where the word starts in the payload (the virtual block ends with the payload: m + 4 <= n)
	if (m + 4 <= 0) return 0;
	uint32_t word = 0;
	if (m >= 0) __builtin_memcpy(&word, payload + m, 4);
	else for (int32_t i = -m; i < 4; ++i) word |= (uint32_t) payload[m + i] << (8 * i);
	if (m < 4) for (int32_t i = 0; i < 4; ++i) if (m + i >= 0 && m + i < 4) word ^= 0xFFu << (8 * i);
	return word;
}
// the CRC-32 of a block from the raw CRCs of the 64 chunks of its virtual block (host stepping of what the lanes of the kernel do together)
inline uint32_t crc32_fold_chunks(const Crc32Tables& t, const uint32_t* chunk /* [CRC32_WAVE_LANES] */) {
	uint32_t group[8];
	for (uint32_t g = 0; g < 8; ++g) { uint32_t c = chunk[8 * g]; for (uint32_t i = 1; i < 8; ++i) c = crc32_apply(t.join[0], c) ^ chunk[8 * g + i]; group[g] = c; }
	uint32_t c = group[0];
	for (uint32_t g = 1; g < 8; ++g) c = crc32_apply(t.join[1], c) ^ group[g];
	return c ^ 0xFFFFFFFFu;
}
inline void crc32_make_tables(Crc32Tables& t) {
	for (uint32_t i = 0; i < 256; ++i) t.slice[0][i] = crc32_table_entry(i);
	for (int k = 1; k < 4; ++k) for (uint32_t i = 0; i < 256; ++i) t.slice[k][i] = (t.slice[k - 1][i] >> 8) ^ t.slice[0][t.slice[k - 1][i] & 0xFFu];
	uint32_t even[32], odd[32];
	odd[0] = CRC32_POLYNOMIAL; // the operator for one zero bit, as in crc32_joined
	uint32_t row = 1;
	for (int k = 1; k < 32; ++k) { odd[k] = row; row <<= 1; }
	gf2_matrix_square(even, odd); gf2_matrix_square(odd, even); // two, four zero bits
	gf2_matrix_square(t.advance[0], odd);                        // eight: one zero byte
	for (int k = 1; k < CRC32_ADVANCE_POWERS; ++k) gf2_matrix_square(t.advance[k], t.advance[k - 1]);
	for (int j = 0; j < 2; ++j) // advance[10]: over 2^10 = CRC32_WAVE_CHUNK zero bytes; advance[13]: over eight chunks
		for (uint32_t b = 0; b < 4; ++b) for (uint32_t v = 0; v < 256; ++v) t.join[j][b][v] = gf2_matrix_times(t.advance[j == 0 ? 10 : 13], v << (8 * b));
}
// crc32(0, bytes, n) of zlib, four bytes per step; `bytes` need not be aligned
AGPU_HD uint32_t crc32_of_sliced(const uint32_t (*slice)[256], const uint8_t* bytes, size_t n) {
	uint32_t c = 0xFFFFFFFFu;
	size_t i = 0;
	for (; i + 4 <= n; i += 4) {
		uint32_t word; __builtin_memcpy(&word, bytes + i, 4);
		c ^= word;
		c = slice[3][c & 0xFFu] ^ slice[2][(c >> 8) & 0xFFu] ^ slice[1][(c >> 16) & 0xFFu] ^ slice[0][c >> 24];
	}
	for (; i < n; ++i) c = slice[0][(c ^ bytes[i]) & 0xFFu] ^ (c >> 8);
	return c ^ 0xFFFFFFFFu;
}
// crc32 of A || B from crc32(A), crc32(B) and |B| < 2^17 with the prepared operators
AGPU_HD uint32_t crc32_joined_with(const uint32_t (*advance)[32], uint32_t crc_a, uint32_t crc_b, uint32_t length_b) {
	for (int k = 0; length_b != 0; length_b >>= 1, ++k) if (length_b & 1u) crc_a = gf2_matrix_times(advance[k], crc_a);
	return crc_a ^ crc_b;
}

}

#endif
