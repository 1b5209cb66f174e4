// tests/emu/words_check.cpp -- TEST ONLY: the word-wise byte helpers of arriba_amd/csrc/device/ingest_core.hpp (find_byte, first_difference, qname_length, same_name,
// compare_names, the three-word load_record) against their byte-by-byte definitions, on random buffers of every length and alignment.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../../arriba_amd/csrc/device/filter_core.hpp"
#include "../../arriba_amd/csrc/device/ingest_core.hpp"
using namespace agpu;

static int failures = 0;
#define CHECK(condition, ...) do { if (!(condition)) { if (++failures < 20) { printf("FAILED %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)

static uint32_t naive_find(const uint8_t* p, uint32_t n, uint8_t byte) { for (uint32_t i = 0; i < n; ++i) if (p[i] == byte) return i; return n; }
static uint32_t naive_difference(const uint8_t* a, const uint8_t* b, uint32_t n) { for (uint32_t i = 0; i < n; ++i) if (a[i] != b[i]) return i; return n; }

// a BAM record with this name (l_read_name = name + NUL [+ padding NULs]), one CIGAR element, no sequence
static std::vector<uint8_t> record_with(const std::string& name, uint32_t padding, int32_t tid, int32_t pos, uint16_t flag) {
	std::vector<uint8_t> r(36 + name.size() + 1 + padding + 4, 0);
	const uint32_t block_size = (uint32_t) r.size() - 4, l_read_name = (uint32_t) name.size() + 1 + padding;
	memcpy(&r[0], &block_size, 4); memcpy(&r[4], &tid, 4); memcpy(&r[8], &pos, 4); r[12] = (uint8_t) l_read_name; r[13] = 7; r[14] = 0x12; r[15] = 0x34;
	const uint16_t n_cigar = 1; memcpy(&r[16], &n_cigar, 2); memcpy(&r[18], &flag, 2);
	const int32_t l_seq = 0; memcpy(&r[20], &l_seq, 4);
	memcpy(&r[36], name.data(), name.size());
	for (uint32_t k = 1; k <= padding; ++k) r[36 + name.size() + k] = (k % 2) ? 'Z' : 0; // (what lies behind the first NUL of the name field must not matter)
	const uint32_t cigar = 50u << 4; memcpy(&r[36 + l_read_name], &cigar, 4);
	return r;
}

// is_tandem_duplication as it stood before the short cut over the first sixteen bases (the restatement of source/read_chimeric_alignments.cpp:215-336, base by base)
static bool is_tandem_duplication_as_written(const Rec* r, uint32_t record, const GenomeView& genome, const uint32_t max_itd_length, TandemAlignment& tandem) {
	const unsigned int min_clipped_length = 12, min_duplication_length = 9, max_duplication_length = max_itd_length;
	const unsigned int max_mismatches = 1, max_non_template_bases = 6, min_alignment_length = 15;
	if (r == nullptr || r->n_cigar == 0) return false;
	unsigned int clipped_length = 0, clipped_position = 0;
	bool clipped_start = true;
	int direction = +1, window_start = 0, window_end = 0, extended_read_start = 0;
	if (r->op(0) == CIGAR_S && r->len(0) >= min_clipped_length) {
		clipped_length = r->len(0);
		clipped_position = 0;
		direction = -1;
		window_start = r->pos + min_duplication_length - clipped_length;
		window_end = r->pos + max_duplication_length - clipped_length;
		extended_read_start = r->pos - clipped_length;
		clipped_start = true;
	}
	if (r->op(r->n_cigar - 1) == CIGAR_S && r->len(r->n_cigar - 1) >= (min_clipped_length > clipped_length ? min_clipped_length : clipped_length)) {
		clipped_length = r->len(r->n_cigar - 1);
		clipped_position = r->l_seq - clipped_length;
		direction = +1;
		window_start = r->endpos() - max_duplication_length;
		window_end = r->endpos() - min_duplication_length;
		extended_read_start = r->endpos();
		clipped_start = false;
	}
	if (clipped_length == 0) return false;
	if (r->contig < 0 || (uint32_t) r->contig >= genome.n_contigs) return false;
	const uint64_t contig_begin = genome.contig_offset[r->contig];
	const uint64_t contig_size = genome.contig_offset[r->contig + 1] - contig_begin; // size_t in the reference
	if (contig_size == 0) return false; // assembly.has()
	const char* contig_sequence = genome.bases + contig_begin;
	if ((uint64_t) (unsigned int) (window_end + max_duplication_length + clipped_length + 1) >= contig_size ||
	    window_start <= (int) (max_duplication_length + clipped_length + 1))
		return false;

	const float min_extended_align_fraction = 0.7;
	unsigned int extended_matches = 0;
	for (unsigned int read_pos = 0; read_pos < clipped_length; ++read_pos)
		if ((uint64_t) (unsigned int) (extended_read_start + read_pos) < contig_size)
			if (contig_sequence[(unsigned int) (extended_read_start + read_pos)] == base_character(r->code((int32_t) (clipped_position + read_pos))))
				extended_matches++;
	if (1.0 * extended_matches / clipped_length >= min_extended_align_fraction) return false;

	for (int contig_pos = window_start; contig_pos <= window_end; ++contig_pos) {
		unsigned int matches = 0, mismatches = 0;
		int tandem_start = (int) contig_size, tandem_end = -1;
		for (unsigned int i = 0; i < clipped_length; i++) {
			const int read_pos = (direction == +1) ? (int) i : (int) (clipped_length - 1 - i);
			if (contig_sequence[contig_pos + read_pos] == base_character(r->code((int32_t) clipped_position + read_pos))) {
				matches++;
				if (contig_pos + read_pos < tandem_start) tandem_start = contig_pos + read_pos;
				if (contig_pos + read_pos > tandem_end) tandem_end = contig_pos + read_pos;
			} else if (i >= max_non_template_bases) {
				mismatches++;
				if (mismatches > max_mismatches) break;
			}
		}
		if (matches >= min_alignment_length || matches + mismatches == clipped_length) {
			tandem.start = tandem_start; tandem.end = tandem_end;
			tandem.record = record;
			tandem.strand = r->forward();
			tandem.first_in_pair = (r->flag & BAMF_READ1) != 0;
			tandem.supplementary = !(r->flag & BAMF_PAIRED) || (clipped_start && r->forward()) || (!clipped_start && !r->forward());
			uint32_t clip_left = clipped_start ? 0 : r->l_seq - clipped_length;
			uint32_t clip_right = clipped_start ? r->l_seq - clipped_length : 0;
			if (tandem_start > contig_pos) clip_left += tandem_start - contig_pos;
			if (tandem_end < contig_pos + (int) clipped_length - 1) clip_right += contig_pos + clipped_length - 1 - tandem_end;
			tandem.n_cigar = 0;
			tandem.cigar[0] = tandem.cigar[1] = tandem.cigar[2] = 0;
			if (clip_left > 0) tandem.cigar[tandem.n_cigar++] = clip_left << 4 | CIGAR_S;
			tandem.cigar[tandem.n_cigar++] = (uint32_t) (tandem_end - tandem_start + 1) << 4 | CIGAR_M;
			if (clip_right > 0) tandem.cigar[tandem.n_cigar++] = clip_right << 4 | CIGAR_S;
			return true;
		}
	}
	return false;
}

// a record with CIGAR elements and bases (codes 1 2 4 8 15 = A C G T N)
static std::vector<uint8_t> record_with_bases(const std::vector<uint32_t>& cigar, const std::vector<uint8_t>& codes, int32_t tid, int32_t pos, uint16_t flag) {
	const std::string name = "r";
	const uint32_t l_read_name = 2, l_seq = (uint32_t) codes.size();
	std::vector<uint8_t> r(36 + l_read_name + 4 * cigar.size() + (l_seq + 1) / 2 + l_seq, 0);
	const uint32_t block_size = (uint32_t) r.size() - 4;
	memcpy(&r[0], &block_size, 4); memcpy(&r[4], &tid, 4); memcpy(&r[8], &pos, 4); r[12] = (uint8_t) l_read_name;
	const uint16_t n_cigar = (uint16_t) cigar.size(); memcpy(&r[16], &n_cigar, 2); memcpy(&r[18], &flag, 2); memcpy(&r[20], &l_seq, 4);
	r[36] = 'r';
	memcpy(&r[36 + l_read_name], cigar.data(), 4 * cigar.size());
	uint8_t* seq = &r[36 + l_read_name + 4 * cigar.size()];
	for (uint32_t i = 0; i < l_seq; ++i) seq[i >> 1] |= (uint8_t) (codes[i] << ((~i & 1) << 2));
	return r;
}
static uint8_t code_of(char base) { return base == 'A' ? 1 : base == 'C' ? 2 : base == 'G' ? 4 : base == 'T' ? 8 : 15; }

// coverage_t::add_fragment one increment per window, as it stood before the ranges of the difference array (source/read_stats.cpp:161-266 restated); `windows` is indexed
// window_offset[contig] + window here, the counts themselves
static void coverage_increment_as_written(uint32_t* window) {
#if defined(__HIP_DEVICE_COMPILE__)
	__hip_atomic_fetch_add(window, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
	__atomic_fetch_add(window, 1u, __ATOMIC_RELAXED);
#endif
}

static void add_fragment_to_coverage_as_written(const CoverageBuild& coverage, const Rec& mate1, uint16_t flag1, const Rec* mate2_or_null, bool is_chimeric) {
	const Rec& mate2 = (mate2_or_null == nullptr) ? mate1 : *mate2_or_null;
	if (coverage.windows == nullptr) return; // (a measurement without coverage_t: ARRIBA_INGEST_SKIP_COVERAGE)
	if (mate1.contig < 0 || (uint32_t) mate1.contig >= coverage.n_contigs || mate2.contig < 0 || (uint32_t) mate2.contig >= coverage.n_contigs) return;
	const uint64_t begin1 = coverage.window_offset[mate1.contig], size1 = coverage.window_offset[mate1.contig + 1] - begin1;
	const uint64_t begin2 = coverage.window_offset[mate2.contig], size2 = coverage.window_offset[mate2.contig + 1] - begin2;
	if (size1 == 0 || size2 == 0) return;
	// the reference compares bam_cigar_type() (0..3) with BAM_CSOFT_CLIP (4), which never matches: only the proper-pair flag turns a fragment chimeric here
	if ((flag1 & BAMF_PAIRED) && !(flag1 & BAMF_PROPER_PAIR)) is_chimeric = true;
	if (!is_chimeric) {
		// (the reference indexes without a check; a position behind its contig would be undefined behaviour there)
		if (!(flag1 & BAMF_REVERSE) || !(flag1 & BAMF_PAIRED)) { const uint64_t w = (uint64_t) (mate1.pos / COVERAGE_RESOLUTION); if (mate1.pos >= 0 && w < size1) coverage.fragment_starts[begin1 + w] = 1; }
		else { const uint64_t w = (uint64_t) (mate2.pos / COVERAGE_RESOLUTION); if (mate2.pos >= 0 && w < size2) coverage.fragment_starts[begin2 + w] = 1; }
	}
	int32_t position1 = mate1.pos, position2 = mate2.pos;
	int32_t position = position1 < position2 ? position1 : position2;
	int32_t window = position / COVERAGE_RESOLUTION;
	uint32_t i1 = 0, i2 = 0;
	while (true) {
		uint32_t op1 = 0, op2 = 0, length1, length2;
		if (i1 < mate1.n_cigar) { op1 = mate1.cigar(i1); length1 = op_consumes_reference(op1 & 15) ? op1 >> 4 : 0; }
		else { length1 = 0; if (position2 / COVERAGE_RESOLUTION > window) window = position2 / COVERAGE_RESOLUTION; }
		if (i2 < mate2.n_cigar) { op2 = mate2.cigar(i2); length2 = op_consumes_reference(op2 & 15) ? op2 >> 4 : 0; }
		else { length2 = 0; if (position1 / COVERAGE_RESOLUTION > window) window = position1 / COVERAGE_RESOLUTION; }
		uint32_t op;
		uint64_t begin, size;
		if (i1 < mate1.n_cigar && (position1 + (int32_t) length1 < position2 + (int32_t) length2 || i2 >= mate2.n_cigar)) {
			i1++;
			if (length1 == 0) continue;
			op = op1; begin = begin1; size = size1; position1 += (int32_t) length1; position = position1;
		} else if (i2 < mate2.n_cigar) {
			i2++;
			if (length2 == 0) continue;
			op = op2; begin = begin2; size = size2; position2 += (int32_t) length2; position = position2;
		} else {
			break;
		}
		if (op_consumes_query(op & 15)) {
			while (window <= position / COVERAGE_RESOLUTION) {
				if (window >= 0 && (uint64_t) window < size && position - window * COVERAGE_RESOLUTION >= COVERAGE_RESOLUTION / 2) coverage_increment_as_written(&coverage.windows[begin + (uint64_t) window]);
				++window;
			}
		} else {
			window = position / COVERAGE_RESOLUTION;
		}
	}
	if (!is_chimeric) {
		if ((flag1 & BAMF_REVERSE) || !(flag1 & BAMF_PAIRED)) { const uint64_t w = (uint64_t) ((position1 - 1) / COVERAGE_RESOLUTION); if (position1 >= 1 && w < size1) coverage.fragment_ends[begin1 + w] = 1; }
		else { const uint64_t w = (uint64_t) ((position2 - 1) / COVERAGE_RESOLUTION); if (position2 >= 1 && w < size2) coverage.fragment_ends[begin2 + w] = 1; }
	}
}


int main() {
	std::mt19937_64 random(20260927);
	// find_byte / first_difference: every length 0..40, every offset 0..8 into the buffer, few distinct byte values so that hits are frequent
	for (int trial = 0; trial < 200000; ++trial) {
		const uint32_t n = random() % 41, offset = random() % 9, alphabet = 1 + random() % 4;
		std::vector<uint8_t> a(offset + n + 16), b(offset + n + 16);
		for (auto& x : a) x = (uint8_t) (random() % alphabet);
		b = a;
		if (n > 0 && random() % 2) b[offset + random() % n] ^= (uint8_t) (1 + random() % 3);
		for (size_t k = offset + n; k < b.size(); ++k) b[k] = (uint8_t) random(); // what lies behind the n bytes must not matter
		const uint8_t byte = (uint8_t) (random() % alphabet);
		CHECK(find_byte(a.data() + offset, n, byte) == naive_find(a.data() + offset, n, byte), "find_byte n=%u offset=%u", n, offset);
		CHECK(first_difference(a.data() + offset, b.data() + offset, n) == naive_difference(a.data() + offset, b.data() + offset, n), "first_difference n=%u offset=%u", n, offset);
	}
	// first_zero_byte: all positions, with bytes of 0x01 and 0x80 around (the classic false positives of the zero-byte trick lie above the first zero only)
	for (int trial = 0; trial < 200000; ++trial) {
		uint64_t v = 0;
		for (int k = 0; k < 8; ++k) { static const uint8_t kinds[] = { 0x00, 0x01, 0x80, 0x7F, 0xFF, 0x2C }; v |= (uint64_t) kinds[random() % 6] << (8 * k); }
		uint32_t expected = 8;
		for (int k = 7; k >= 0; --k) if (((v >> (8 * k)) & 0xFF) == 0) expected = (uint32_t) k;
		CHECK(first_zero_byte(v) == expected, "first_zero_byte %016llx", (unsigned long long) v);
	}
	// records: load_record's fields, qname_length, same_name, compare_names against std::string
	for (int trial = 0; trial < 100000; ++trial) {
		auto make_name = [&](uint32_t length) { std::string s; for (uint32_t k = 0; k < length; ++k) s += (char) ("AB,:/0"[random() % 6]); return s; };
		std::string x = make_name(1 + random() % 40), y = random() % 3 == 0 ? x : make_name(1 + random() % 40);
		if (random() % 4 == 0) y = x.substr(0, 1 + random() % x.size()); // a prefix
		const int32_t tid_x = (int32_t) (random() % 5) - 1, pos_x = (int32_t) (random() % 1000000) - 1; const uint16_t flag_x = (uint16_t) random();
		const std::vector<uint8_t> rx = record_with(x, random() % 12, tid_x, pos_x, flag_x), ry = record_with(y, random() % 12, 1, 5, 0);
		std::vector<uint8_t> stream(rx); stream.insert(stream.end(), ry.begin(), ry.end()); stream.resize(stream.size() + 16, 0xEE);
		const uint64_t offsets[2] = { 0, rx.size() };
		const uint32_t tid_to_contig[4] = { 10, 11, 12, 13 };
		IngestStream in; in.bytes = stream.data(); in.size = stream.size(); in.record_offset = offsets; in.n_records = 2; in.n_targets = 4; in.tid_to_contig = tid_to_contig; in.hit_index = nullptr;
		const Rec a = load_record(in, 0), b = load_record(in, 1);
		CHECK(a.pos == pos_x && a.flag == flag_x && a.n_cigar == 1 && a.l_seq == 0 && a.contig == (tid_x >= 0 ? 10 + tid_x : -1) && a.cigar(0) == (50u << 4), "load_record");
		CHECK(qname_length(a) == x.size() && qname_length(b) == y.size(), "qname_length %zu %zu", x.size(), y.size());
		const int64_t hi_x = 1 + (int64_t) (random() % 3), hi_y = 1 + (int64_t) (random() % 3);
		CHECK(same_name(a, hi_x, b, hi_y) == (x == y && hi_x == hi_y), "same_name");
		for (int itd_x = 0; itd_x < 2; ++itd_x) for (int itd_y = 0; itd_y < 2; ++itd_y) {
			const std::string full_x = x + "," + std::to_string(hi_x) + (itd_x ? "ITD" : ""), full_y = y + "," + std::to_string(hi_y) + (itd_y ? "ITD" : "");
			const int expected = full_x.compare(full_y) < 0 ? -1 : full_x.compare(full_y) > 0 ? 1 : 0;
			CHECK(compare_names(fragment_name(a, hi_x, itd_x), fragment_name(b, hi_y, itd_y, (uint32_t) y.size())) == expected, "compare_names '%s' '%s'", full_x.c_str(), full_y.c_str());
		}
	}
	// is_tandem_duplication: the short cut against the loop as written -- random contigs (a small alphabet, so that chance hits happen), clipped reads whose clipped bases are
	// random, a copy of the contig nearby (a real tandem duplication) or such a copy with a few changed bases; clips of 12..40 bases at either end; both strands
	{
		uint64_t hits = 0, calls = 0;
		for (int trial = 0; trial < 60000; ++trial) {
			const uint32_t contig_length = 2000, alphabet = 2 + random() % 3;
			std::string contig(contig_length, 'A');
			for (auto& c : contig) c = "ACGTN"[random() % alphabet];
			const uint32_t max_itd_length = 9 + random() % 120;
			const uint32_t clip = 12 + random() % 29, aligned = 30 + random() % 40;
			const bool clipped_start = random() % 2;
			const int32_t pos = 400 + (int32_t) (random() % 1000);
			std::vector<uint32_t> cigar;
			if (clipped_start) cigar.push_back(clip << 4 | CIGAR_S);
			cigar.push_back(aligned << 4 | CIGAR_M);
			if (!clipped_start) cigar.push_back(clip << 4 | CIGAR_S);
			if (random() % 8 == 0) cigar.insert(cigar.begin() + (clipped_start ? 1 : 0), (uint32_t) (5u << 4 | CIGAR_M)), cigar.insert(cigar.begin() + (clipped_start ? 2 : 1), (uint32_t) (50u << 4 | CIGAR_N));
			std::vector<uint8_t> codes(clip + aligned + (cigar.size() > 3 ? 5 : 0));
			for (auto& c : codes) c = code_of("ACGTN"[random() % alphabet]);
			// the clipped bases: random, or the contig from somewhere inside the window the function searches, with 0-3 bases changed
			const uint32_t kind = random() % 3;
			if (kind > 0) {
				const int32_t end = pos + (int32_t) aligned + (cigar.size() > 3 ? 55 : 0);
				const int32_t from = clipped_start ? pos + 9 - (int32_t) clip + (int32_t) (random() % (max_itd_length + 1)) : end - (int32_t) max_itd_length + (int32_t) (random() % (max_itd_length + 1));
				for (uint32_t k = 0; k < clip; ++k) { const int32_t at = from + (int32_t) k; if (at >= 0 && at < (int32_t) contig_length) codes[(clipped_start ? 0 : codes.size() - clip) + k] = code_of(contig[at]); }
				for (uint32_t changes = kind == 2 ? random() % 4 : 0; changes > 0; --changes) codes[(clipped_start ? 0 : codes.size() - clip) + random() % clip] = code_of("ACGT"[random() % 4]);
			}
			const uint16_t flag = (uint16_t) ((random() % 2 ? BAMF_REVERSE : 0) | (random() % 2 ? BAMF_PAIRED | BAMF_PROPER_PAIR : 0) | (random() % 2 ? BAMF_READ1 : 0));
			std::vector<uint8_t> stream = record_with_bases(cigar, codes, 0, pos, flag);
			stream.resize(stream.size() + 16, 0);
			const uint64_t offsets[1] = { 0 }; const uint32_t tid_to_contig[1] = { 0 };
			IngestStream in; in.bytes = stream.data(); in.size = stream.size(); in.record_offset = offsets; in.n_records = 1; in.n_targets = 1; in.tid_to_contig = tid_to_contig; in.hit_index = nullptr;
			const Rec r = load_record(in, 0);
			const uint64_t contig_offset[2] = { 0, contig_length }; const uint8_t contig_bits[1] = { 0 };
			GenomeView genome; genome.n_contigs = 1; genome.contig_offset = contig_offset; genome.contig_bits = contig_bits; genome.bases = contig.data();
			TandemAlignment x, y; memset(&x, 0, sizeof(x)); memset(&y, 0, sizeof(y));
			const bool found_x = is_tandem_duplication(&r, 7, genome, max_itd_length, x), found_y = is_tandem_duplication_as_written(&r, 7, genome, max_itd_length, y);
			++calls; hits += found_y;
			CHECK(found_x == found_y && (!found_y || memcmp(&x, &y, sizeof(x)) == 0), "is_tandem_duplication: clip %u at the %s, max_itd_length %u: %d vs %d", clip, clipped_start ? "start" : "end", max_itd_length, (int) found_x, (int) found_y);
		}
		CHECK(hits > calls / 20 && hits < calls - calls / 20, "is_tandem_duplication was to be tried on hits and misses alike: %llu hits of %llu", (unsigned long long) hits, (unsigned long long) calls);
		printf("is_tandem_duplication: %llu hits of %llu calls equal\n", (unsigned long long) hits, (unsigned long long) calls);
	}
	// add_fragment_to_coverage: ranges in the difference array + prefix sums against one increment per window -- pairs and single mates, mates that overlap, lie apart or on
	// two contigs, deletions and introns (a deletion makes the reference count a window twice), insertions, clips, reads at the first and behind the last window of a contig,
	// positions of -1, contigs without windows
	{
		uint64_t fragments = 0, increments = 0;
		for (int trial = 0; trial < 3000; ++trial) {
			const uint32_t n_contigs = 1 + random() % 4;
			std::vector<uint64_t> window_offset(n_contigs + 1, 0);
			std::vector<uint32_t> contig_length(n_contigs);
			for (uint32_t c = 0; c < n_contigs; ++c) { contig_length[c] = random() % 5 == 0 ? 0 : 200 + (uint32_t) (random() % 3000); window_offset[c + 1] = window_offset[c] + (contig_length[c] ? contig_length[c] / COVERAGE_RESOLUTION + 1 : 0); }
			const uint64_t windows = window_offset[n_contigs];
			std::vector<uint32_t> expected(windows + 1, 0), differences(coverage_difference_slots(windows, n_contigs) + 1, 0);
			std::vector<uint8_t> starts_a(windows + 1, 0), ends_a(windows + 1, 0), starts_b(windows + 1, 0), ends_b(windows + 1, 0);
			CoverageBuild as_written = { n_contigs, window_offset.data(), expected.data(), starts_a.data(), ends_a.data() }, ranges = { n_contigs, window_offset.data(), differences.data(), starts_b.data(), ends_b.data() };
			for (int fragment = 0; fragment < 40; ++fragment) {
				auto make_cigar = [&](std::vector<uint32_t>& cigar, uint32_t& query) {
					query = 0;
					if (random() % 3 == 0) { const uint32_t l = 1 + random() % 30; cigar.push_back(l << 4 | CIGAR_S); query += l; }
					const uint32_t blocks = 1 + random() % 4;
					for (uint32_t k = 0; k < blocks; ++k) {
						if (k > 0) { const uint32_t kind = random() % 3; if (kind == 0) cigar.push_back((uint32_t) (1 + random() % 25) << 4 | CIGAR_D); else if (kind == 1) cigar.push_back((uint32_t) (1 + random() % 400) << 4 | CIGAR_N); else { const uint32_t l = 1 + random() % 5; cigar.push_back(l << 4 | CIGAR_I); query += l; } }
						const uint32_t l = 1 + random() % 90; cigar.push_back(l << 4 | (random() % 6 == 0 ? CIGAR_X : CIGAR_M)); query += l;
					}
					if (random() % 3 == 0) { const uint32_t l = 1 + random() % 30; cigar.push_back(l << 4 | (random() % 4 == 0 ? CIGAR_H : CIGAR_S)); if ((cigar.back() & 15) == CIGAR_S) query += l; }
				};
				const int32_t tid1 = (int32_t) (random() % n_contigs), tid2 = random() % 6 == 0 ? (int32_t) (random() % n_contigs) : tid1;
				const int32_t span = (int32_t) contig_length[tid1] + 400;
				const int32_t pos1 = random() % 25 == 0 ? -1 : (int32_t) (random() % span) - (random() % 10 == 0 ? 0 : 0), pos2 = random() % 3 == 0 ? pos1 + (int32_t) (random() % 40) - 10 : (int32_t) (random() % ((int32_t) contig_length[tid2] + 400));
				std::vector<uint32_t> cigar1, cigar2; uint32_t query1, query2;
				make_cigar(cigar1, query1); make_cigar(cigar2, query2);
				const bool pair = random() % 3 != 0;
				uint16_t flag = (uint16_t) ((random() % 2 ? BAMF_REVERSE : 0) | (pair || random() % 2 ? BAMF_PAIRED : 0) | (random() % 4 ? BAMF_PROPER_PAIR : 0));
				std::vector<uint8_t> stream = record_with_bases(cigar1, std::vector<uint8_t>(query1, 1), tid1, pos1, flag);
				const uint64_t second = stream.size();
				const std::vector<uint8_t> other = record_with_bases(cigar2, std::vector<uint8_t>(query2, 1), tid2, pos2 < -1 ? -1 : pos2, (uint16_t) (flag ^ BAMF_REVERSE));
				stream.insert(stream.end(), other.begin(), other.end()); stream.resize(stream.size() + 16, 0);
				const uint64_t offsets[2] = { 0, second }; std::vector<uint32_t> tid_to_contig(n_contigs); for (uint32_t c = 0; c < n_contigs; ++c) tid_to_contig[c] = c;
				IngestStream in; in.bytes = stream.data(); in.size = stream.size(); in.record_offset = offsets; in.n_records = 2; in.n_targets = n_contigs; in.tid_to_contig = tid_to_contig.data(); in.hit_index = nullptr;
				const Rec mate1 = load_record(in, 0), mate2 = load_record(in, 1);
				const bool is_chimeric = random() % 2;
				const uint16_t flag_passed = random() % 5 == 0 ? 0 : flag;
				add_fragment_to_coverage_as_written(as_written, mate1, flag_passed, pair ? &mate2 : nullptr, is_chimeric);
				add_fragment_to_coverage(ranges, mate1, flag_passed, pair ? &mate2 : nullptr, is_chimeric);
				++fragments;
			}
			// as coverage_from_differences_kernel: window w of contig c is slot window_offset[c] + c + w; every contig sums to zero
			uint32_t running = 0; uint64_t slot = 0; bool equal = true, closed = true;
			for (uint32_t c = 0; c < n_contigs; ++c) {
				for (uint64_t w = window_offset[c]; w < window_offset[c + 1]; ++w) { running += differences[slot++]; equal = equal && running == expected[w]; increments += expected[w]; }
				running += differences[slot++]; closed = closed && running == 0;
			}
			CHECK(equal && closed, "add_fragment_to_coverage: the windows from the ranges are not those of the increments (trial %d)", trial);
			CHECK(starts_a == starts_b && ends_a == ends_b, "add_fragment_to_coverage: start / end flags (trial %d)", trial);
		}
		CHECK(increments > 20 * fragments / 10, "add_fragment_to_coverage was to be tried on fragments that cover windows: %llu increments of %llu fragments", (unsigned long long) increments, (unsigned long long) fragments);
		printf("add_fragment_to_coverage: %llu fragments, %llu increments equal\n", (unsigned long long) fragments, (unsigned long long) increments);
	}
	// the hit index kept beside a record (record_parse_kernel: hit_index_to_keep) and read back (hit_index_of) against the walk over the aux fields, for every integer type of
	// the HI field, values that do not fit 32 bits or are the marker itself, records without HI, HI behind other fields
	{
		const int64_t values[] = { 0, 1, 2, 127, -128, 255, 32767, -32768, 65535, 2147483647LL, -2147483647LL - 1, 4294967295LL, 3000000000LL, -1 };
		for (int64_t value : values) for (char type : std::string("cCsSiI")) for (int position = 0; position < 3; ++position) {
			const bool fits = (type == 'c' && value >= -128 && value <= 127) || (type == 'C' && value >= 0 && value <= 255) || (type == 's' && value >= -32768 && value <= 32767) ||
			                  (type == 'S' && value >= 0 && value <= 65535) || (type == 'i' && value >= -2147483647LL - 1 && value <= 2147483647LL) || (type == 'I' && value >= 0 && value <= 4294967295LL);
			if (!fits) continue;
			std::vector<uint8_t> r = record_with("name", 0, 0, 10, 0);
			auto tag = [&r](const char* name, char kind, int64_t v) { r.push_back(name[0]); r.push_back(name[1]); r.push_back((uint8_t) kind); const int width = (kind == 'c' || kind == 'C') ? 1 : (kind == 's' || kind == 'S') ? 2 : 4; for (int b = 0; b < width; ++b) r.push_back((uint8_t) ((uint64_t) v >> (8 * b))); };
			if (position >= 1) tag("NH", 'C', 3);
			if (position == 2) { r.push_back('S'); r.push_back('A'); r.push_back('Z'); for (char c : std::string("1,100,+,50M,60,0;")) r.push_back((uint8_t) c); r.push_back(0); }
			tag("HI", type, value);
			tag("AS", 'C', 98);
			const uint32_t block_size = (uint32_t) r.size() - 4; memcpy(&r[0], &block_size, 4);
			r.resize(r.size() + 16, 0);
			const uint64_t offsets[1] = { 0 }; const uint32_t tid_to_contig[1] = { 0 };
			IngestStream in; in.bytes = r.data(); in.size = r.size(); in.record_offset = offsets; in.n_records = 1; in.n_targets = 1; in.tid_to_contig = tid_to_contig; in.hit_index = nullptr;
			const Rec record = load_record(in, 0);
			const AuxTags tags = scan_aux(record.aux, record.end);
			CHECK(tags.has_hi && tags.hi == value && tags.has_sa == (position == 2), "scan_aux: HI:%c:%lld", type, (long long) value);
			const int32_t kept[1] = { hit_index_to_keep(tags) };
			CHECK((kept[0] == HIT_INDEX_UNKNOWN) == (value > 2147483647LL || value <= -2147483647LL - 1), "hit_index_to_keep: HI:%c:%lld kept as %d", type, (long long) value, kept[0]);
			in.hit_index = kept;
			CHECK(hit_index_of(in, 0, record) == value, "hit_index_of: HI:%c:%lld", type, (long long) value);
		}
		std::vector<uint8_t> r = record_with("name", 0, 0, 10, 0); r.resize(r.size() + 16, 0);
		const uint64_t offsets[1] = { 0 }; const uint32_t tid_to_contig[1] = { 0 };
		IngestStream in; in.bytes = r.data(); in.size = r.size() - 16; in.record_offset = offsets; in.n_records = 1; in.n_targets = 1; in.tid_to_contig = tid_to_contig; in.hit_index = nullptr;
		const Rec record = load_record(in, 0);
		const AuxTags none = scan_aux(record.aux, record.end);
		const int32_t kept[1] = { hit_index_to_keep(none) };
		in.hit_index = kept;
		CHECK(!none.has_hi && kept[0] == 1 && hit_index_of(in, 0, record) == 1, "a record without HI has hit index 1");
	}
	printf(failures ? "words_check: %d FAILED\n" : "words_check: ok\n", failures);
	return failures != 0;
}
