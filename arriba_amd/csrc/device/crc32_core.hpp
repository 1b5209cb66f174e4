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

}

#endif
