// arriba_amd/csrc/device/ingest_core.hpp -- read_chimeric_alignments on the device: BAM records in HBM -> chimeric fragments.
//
// The stream of uncompressed BAM records lies in HBM as it came out of the container.  The functions below restate, on the raw record bytes,
//   the record loop of read_chimeric_alignments   reference: source/read_chimeric_alignments.cpp:585-757   (replay_group)
//   add_chimeric_alignment                        reference: source/read_chimeric_alignments.cpp:50-91     (materialize)
//   find_spanning_intron                          reference: source/read_chimeric_alignments.cpp:19-41
//   extract_read_through_alignment                reference: source/read_chimeric_alignments.cpp:93-193
//   clipped_sequence_is_adapter                   reference: source/read_chimeric_alignments.cpp:197-211
//   is_tandem_duplication                         reference: source/read_chimeric_alignments.cpp:215-336
//   disjoin_split_read_segments                   reference: source/read_chimeric_alignments.cpp:340-373
//   remove_malformed_alignments                   reference: source/read_chimeric_alignments.cpp:377-506   (normalize_plan)
//   is_clipped_at_correct_end / is_pristine_alignment   reference: source/read_chimeric_alignments.cpp:511-558
//   coverage_t::add_fragment                      reference: source/read_stats.cpp:161-266
// What a record does depends only on the records of the same read name ("QNAME,HI") in front of it (the parked first mate, the alignment list of the
// name) and on read-only data; counters and coverage windows commute.  So the records are grouped by name (hash + stable sort keeps the file order
// inside a group) and one thread replays the reference's loop body over the records of one name.  A fragment is kept as a PLAN (which record, which
// part of its CIGAR); start / end / CIGAR / sequence are derived from the record bytes whenever they are needed, nothing is copied before the final pack.
// __host__ __device__: tests/emu steps the identical code on a CPU-only box; the product runs it inside the kernels of agpu_ingest.hip only.
#ifndef AGPU_INGEST_CORE_HPP
#define AGPU_INGEST_CORE_HPP 1

#include "annotate_core.hpp"
#include "event_core.hpp"

namespace agpu {

// ---- raw record access (SAMv1 section 4.2; little endian; fields are not aligned) ------------------------------------------------------------------

AGPU_HD uint32_t load_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
AGPU_HD uint64_t load_u64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
// Bytes of the stream eight at a time.  A lane that walks a read name byte by byte issues one load per byte, and the 64 lanes of a wavefront stand in 64 different records: 64
// cache lines looked up per byte.  The names of a group were walked ~550 bytes worth per read name (lengths, comparisons, the comma) -- half of the replay kernel's loads.
// index of the first zero byte of v (byte 0 = the lowest address), 8 if it has none
AGPU_HD uint32_t first_zero_byte(uint64_t v) { const uint64_t z = (v - 0x0101010101010101ull) & ~v & 0x8080808080808080ull; return z ? (uint32_t) __builtin_ctzll(z) >> 3 : 8u; } // (bits above the first zero byte may be wrong, the lowest one never is)
AGPU_HD uint64_t low_bytes(uint32_t n) { return n >= 8 ? ~0ull : (1ull << (8 * n)) - 1; } // 0xFF in the n lowest bytes
// index of the first byte equal to `byte` among the n bytes at p, n if there is none; p[0..n) readable
AGPU_HD uint32_t find_byte(const uint8_t* p, uint32_t n, uint8_t byte) {
	const uint64_t pattern = 0x0101010101010101ull * byte;
	uint32_t i = 0;
	for (; i + 8 <= n; i += 8) { const uint32_t z = first_zero_byte(load_u64(p + i) ^ pattern); if (z < 8) return i + z; }
	if (i == n) return n;
	if (n >= 8) { // the last word again, the bytes seen before made non-zero
		const uint32_t z = first_zero_byte((load_u64(p + n - 8) ^ pattern) | low_bytes(i - (n - 8)));
		return z < 8 ? n - 8 + z : n;
	}
	for (; i < n; ++i) if (p[i] == byte) return i;
	return n;
}
// index of the first byte in which a[0..n) and b[0..n) differ, n if they are equal
AGPU_HD uint32_t first_difference(const uint8_t* a, const uint8_t* b, uint32_t n) {
	uint32_t i = 0;
	for (; i + 8 <= n; i += 8) { const uint64_t x = load_u64(a + i) ^ load_u64(b + i); if (x) return i + ((uint32_t) __builtin_ctzll(x) >> 3); }
	if (i == n) return n;
	if (n >= 8) {
		const uint64_t x = (load_u64(a + n - 8) ^ load_u64(b + n - 8)) & ~low_bytes(i - (n - 8));
		return x ? n - 8 + ((uint32_t) __builtin_ctzll(x) >> 3) : n;
	}
	for (; i < n; ++i) if (a[i] != b[i]) return i;
	return n;
}
AGPU_HD uint16_t load_u16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }

enum : uint16_t { BAMF_PAIRED = 1, BAMF_PROPER_PAIR = 2, BAMF_UNMAP = 4, BAMF_MUNMAP = 8, BAMF_REVERSE = 16, BAMF_READ1 = 64, BAMF_SECONDARY = 256, BAMF_DUP = 1024, BAMF_SUPPLEMENTARY = 2048 };

AGPU_HD bool op_consumes_query(uint32_t op) { return (0x3C1A7u >> (op << 1)) & 1; }      // bam_cigar_type, htslib
AGPU_HD bool op_consumes_reference(uint32_t op) { return (0x3C1A7u >> (op << 1)) & 2; }

struct IngestStream {
	const uint8_t* bytes;            // uncompressed BAM stream (header + records)
	uint64_t size;
	const uint64_t* record_offset;   // [n_records] offset of the block_size word of every record
	uint64_t n_records;
	uint32_t n_targets;
	const uint32_t* tid_to_contig;   // BAM reference id -> contig id of the session
	const int32_t* hit_index;        // [n_records] the HI of every active record as record_parse_kernel found it (1 without one), HIT_INDEX_UNKNOWN = look again; may be null
};
const int32_t HIT_INDEX_UNKNOWN = -0x7FFFFFFF - 1;

// one record, its fixed fields decoded (the variable parts stay in the stream)
struct Rec {
	const uint8_t* cigar_bytes; const uint8_t* seq_bytes; const uint8_t* name; const uint8_t* aux; const uint8_t* end;
	int32_t contig, pos, l_seq;
	uint32_t n_cigar, l_read_name;
	uint16_t flag;
	AGPU_HD uint32_t cigar(uint32_t i) const { return load_u32(cigar_bytes + 4 * (size_t) i); }
	AGPU_HD uint32_t op(uint32_t i) const { return cigar(i) & 15; }
	AGPU_HD uint32_t len(uint32_t i) const { return cigar(i) >> 4; }
	AGPU_HD uint32_t code(int32_t i) const { return (seq_bytes[i >> 1] >> ((~i & 1) << 2)) & 15; } // 4-bit base code, "=ACMGRSVTWYHKDBN"
	AGPU_HD bool forward() const { return !(flag & BAMF_REVERSE); }
	AGPU_HD bool paired() const { return flag & BAMF_PAIRED; }
	AGPU_HD int32_t qlen(uint32_t n) const { int32_t l = 0; for (uint32_t i = 0; i < n; ++i) if (op_consumes_query(op(i))) l += (int32_t) len(i); return l; }
	AGPU_HD int32_t rlen(uint32_t n) const { int32_t l = 0; for (uint32_t i = 0; i < n; ++i) if (op_consumes_reference(op(i))) l += (int32_t) len(i); return l; }
	AGPU_HD int32_t endpos() const { int32_t l = (!(flag & BAMF_UNMAP) && n_cigar > 0) ? rlen(n_cigar) : 1; return pos + (l == 0 ? 1 : l); } // bam_endpos
};

// validity of the sizes inside a record block (checked once per record; a violation is the reference's "failed to load alignments")
AGPU_HD bool record_sizes_ok(const uint8_t* block, uint32_t block_size) {
	if (block_size < 32) return false;
	const uint32_t l_read_name = block[8], n_cigar = load_u16(block + 12);
	const int32_t l_seq = (int32_t) load_u32(block + 16);
	if (l_seq < 0 || l_read_name == 0) return false;
	const uint64_t fixed = 32ull + l_read_name + 4ull * n_cigar + ((uint64_t) l_seq + 1) / 2 + (uint64_t) l_seq;
	return fixed <= block_size;
}

// (p: where the record starts -- in the stream, or in a copy of a piece of it: record_parse_kernel stages the records of a wavefront in LDS)
AGPU_HD Rec load_record_at(const IngestStream& in, const uint8_t* p) {
	const uint64_t word0 = load_u64(p), word1 = load_u64(p + 8), word2 = load_u64(p + 16); // block_size, refID | pos, l_read_name mapq bin | n_cigar, flag, l_seq: three loads instead of seven
	const uint32_t block_size = (uint32_t) word0;
	p += 4;
	Rec rec;
	const int32_t tid = (int32_t) (word0 >> 32);
	rec.contig = (tid >= 0 && (uint32_t) tid < in.n_targets) ? (int32_t) in.tid_to_contig[tid] : -1;
	rec.pos = (int32_t) (uint32_t) word1;
	rec.l_read_name = (uint32_t) (word1 >> 32) & 0xFFu;
	rec.n_cigar = (uint32_t) word2 & 0xFFFFu;
	rec.flag = (uint16_t) (word2 >> 16);
	rec.l_seq = (int32_t) (word2 >> 32);
	rec.name = p + 32;
	rec.cigar_bytes = rec.name + rec.l_read_name;
	rec.seq_bytes = rec.cigar_bytes + 4 * (size_t) rec.n_cigar;
	rec.aux = rec.seq_bytes + ((size_t) rec.l_seq + 1) / 2 + (size_t) rec.l_seq;
	rec.end = p + block_size;
	return rec;
}
AGPU_HD Rec load_record(const IngestStream& in, uint32_t r) { return load_record_at(in, in.bytes + in.record_offset[r]); }

// ---- the record chain (every record starts where the one before it ends), cut in parallel ---------------------------------------------------------
// Every 8 KB segment of the stream guesses its first record (two consecutive plausible headers) and walks the chain to its end; a guess counts only if it is the
// exact end of the segment before it; runs of segments where it is not are walked again from the last good end.  By induction from the known first record the
// result is the true chain, whatever the guesses were.

const uint32_t SEGMENT_BYTES = 8192;
const uint64_t SEARCH_LIMIT = 4u << 20;    // how far a segment looks for its first record before it leaves the answer to the repair pass
const uint64_t OFFSET_NONE = ~0ull, OFFSET_BROKEN = ~0ull - 1;

// a record header that could be real: sizes consistent, reference ids in range, the name terminated
AGPU_HD bool plausible_record(const uint8_t* bytes, uint64_t size, uint64_t o, uint32_t n_targets) {
	if (o + 36 > size) return false;
	const uint32_t block_size = load_u32(bytes + o);
	if (block_size < 32 || block_size > (1u << 28) || o + 4 + (uint64_t) block_size > size) return false;
	const uint8_t* p = bytes + o + 4;
	const int32_t tid = (int32_t) load_u32(p), pos = (int32_t) load_u32(p + 4), next_tid = (int32_t) load_u32(p + 20), next_pos = (int32_t) load_u32(p + 24);
	if (tid < -1 || tid >= (int32_t) n_targets || next_tid < -1 || next_tid >= (int32_t) n_targets || pos < -1 || next_pos < -1) return false;
	if (!record_sizes_ok(p, block_size)) return false;
	const uint32_t l_read_name = p[8];
	return p[32 + l_read_name - 1] == 0;
}

// walks the chain from `start` to the first record at or behind `segment_end`; OFFSET_BROKEN if a block size cannot be right
AGPU_HD uint64_t walk_segment(const uint8_t* bytes, uint64_t size, uint64_t start, uint64_t segment_end, uint32_t& count, uint64_t* offsets) {
	uint64_t o = start;
	count = 0;
	while (o < segment_end) {
		if (o + 4 > size) return OFFSET_BROKEN;
		const uint32_t block_size = load_u32(bytes + o);
		if (block_size < 32 || o + 4 + (uint64_t) block_size > size) return OFFSET_BROKEN;
		if (offsets != nullptr) offsets[count] = o;
		++count;
		o += 4 + (uint64_t) block_size;
	}
	return o;
}

AGPU_HD void guess_segment(const uint8_t* bytes, uint64_t size, uint64_t base, uint64_t s, uint32_t n_targets, uint64_t* first, uint64_t* end, uint32_t* count) {
	const uint64_t begin = base + s * SEGMENT_BYTES, segment_end = (begin + SEGMENT_BYTES < size) ? begin + SEGMENT_BYTES : size;
	uint64_t start = OFFSET_NONE;
	if (s == 0) start = base;
	else {
		const uint64_t limit = (begin + SEARCH_LIMIT < size) ? begin + SEARCH_LIMIT : size;
		for (uint64_t o = begin; o < limit; ++o) {
			if (!plausible_record(bytes, size, o, n_targets)) continue;
			const uint64_t next = o + 4 + (uint64_t) load_u32(bytes + o);
			if (next == size || plausible_record(bytes, size, next, n_targets)) { start = o; break; }
		}
	}
	first[s] = start;
	uint32_t n = 0;
	end[s] = (start == OFFSET_NONE) ? OFFSET_NONE : walk_segment(bytes, size, start, segment_end, n, nullptr);
	count[s] = n;
}

// A run of segments that do not start where the segment before them ends is walked again by ONE caller, from the end of the last good segment in front of the run
// through the whole run (`previous_end`: the ends as they were before this pass).  Only the first segment of a run does the work: a segment behind a wrong one is
// flagged, too, although its own guess may be right -- repairing it from its neighbour's wrong end would push the error one segment further with every pass (a false
// start two bytes in front of a record, once in ~3*10^5 segments of the bench's stream, took 1865 passes that way).
AGPU_HD void repair_segment_run(const uint8_t* bytes, uint64_t size, uint64_t base, uint64_t n_segments, uint64_t s, const uint8_t* mismatch, const uint64_t* previous_end, uint64_t* first, uint64_t* end, uint32_t* count) {
	if (!mismatch[s] || mismatch[s - 1]) return; // (s >= 1 for every flagged segment)
	uint64_t start = previous_end[s - 1];
	for (uint64_t t = s; t < n_segments && (t == s || mismatch[t]); ++t) {
		const uint64_t begin = base + t * SEGMENT_BYTES, segment_end = (begin + SEGMENT_BYTES < size) ? begin + SEGMENT_BYTES : size;
		first[t] = start;
		uint32_t n = 0;
		end[t] = (start >= OFFSET_BROKEN) ? start : walk_segment(bytes, size, start, segment_end, n, nullptr);
		count[t] = n;
		start = end[t];
	}
}

// aux fields: the first "HI" (integer types) and the presence of "SA", as bam_aux_get would find them (a malformed field ends the walk)
struct AuxTags { bool has_hi, has_sa; int64_t hi; };
AGPU_HD AuxTags scan_aux(const uint8_t* s, const uint8_t* end) {
	AuxTags tags; tags.has_hi = false; tags.has_sa = false; tags.hi = 0;
	while (s + 3 <= end) {
		const uint8_t* value = s + 2;
		size_t size;
		switch (*value) {
			case 'A': case 'c': case 'C': size = 2; break;
			case 's': case 'S': size = 3; break;
			case 'i': case 'I': case 'f': size = 5; break;
			case 'd': size = 9; break;
			case 'Z': case 'H': { const uint8_t* p = value + 1; while (p < end && *p) ++p; if (p >= end) return tags; size = (size_t) (p - value) + 1; break; }
			case 'B': {
				if (value + 6 > end) return tags;
				const size_t element = (value[1] == 'c' || value[1] == 'C') ? 1 : (value[1] == 's' || value[1] == 'S') ? 2 : 4;
				size = 6 + element * (size_t) load_u32(value + 2);
				break;
			}
			default: return tags;
		}
		if (value + size > end) return tags;
		if (s[0] == 'H' && s[1] == 'I' && !tags.has_hi) {
			tags.has_hi = true;
			switch (*value) { // bam_aux2i
				case 'c': tags.hi = (int8_t) value[1]; break;
				case 'C': tags.hi = value[1]; break;
				case 's': tags.hi = (int16_t) load_u16(value + 1); break;
				case 'S': tags.hi = load_u16(value + 1); break;
				case 'i': tags.hi = (int32_t) load_u32(value + 1); break;
				case 'I': tags.hi = load_u32(value + 1); break;
				default: tags.hi = 0; break;
			}
		}
		if (s[0] == 'S' && s[1] == 'A') tags.has_sa = true;
		if (tags.has_hi && tags.has_sa) return tags;
		s = value + size;
	}
	return tags;
}

// "HI" of a record, 1 without one (source/read_chimeric_alignments.cpp:618-625): the aux fields are walked once, by record_parse_kernel; whoever needs the hit index of the
// record later -- the name of its fragment is "QNAME,HI" -- finds it beside the record's status instead of walking the fields again (five tags in front of it in STAR's
// records, every one a dependent load)
AGPU_HD int32_t hit_index_to_keep(const AuxTags& tags) { const int64_t hi = tags.has_hi ? tags.hi : 1; return (hi > HIT_INDEX_UNKNOWN && hi <= 0x7FFFFFFF) ? (int32_t) hi : HIT_INDEX_UNKNOWN; }
AGPU_HD int64_t hit_index_of(const IngestStream& in, uint32_t record, const Rec& rec) {
	if (in.hit_index != nullptr) { const int32_t kept = in.hit_index[record]; if (kept != HIT_INDEX_UNKNOWN) return kept; }
	const AuxTags tags = scan_aux(rec.aux, rec.end);
	return tags.has_hi ? tags.hi : 1;
}

// per-record verdict of the first lines of the loop body (source/read_chimeric_alignments.cpp:611-631)
enum : uint8_t { RECORD_SKIPPED = 0, RECORD_ACTIVE = 1, RECORD_MISSING_HI = 2, RECORD_BROKEN = 3, RECORD_STATUS_MASK = 3, RECORD_HAS_SA = 4 };

AGPU_HD uint32_t qname_length(const Rec& r) { return find_byte(r.name, r.l_read_name, 0); } // (the name up to its NUL, at most l_read_name bytes)

// One sample over several contexts (agpu_shard_merge): no read name may have records in two parts.  Every run of records with one QNAME in the stream of a part gives
// a 128-bit key of the QNAME (two FNV-1a passes with different offsets, finalised); equal keys anywhere in the sample -- two runs of a name in one part or in two --
// mean that the alignments of a read do not follow each other in the file.
AGPU_HD bool starts_qname_run(const IngestStream& in, uint32_t record) {
	if (record == 0) return true;
	const Rec a = load_record(in, record - 1), b = load_record(in, record);
	const uint32_t length = qname_length(b);
	if (qname_length(a) != length) return true;
	for (uint32_t i = 0; i < length; ++i) if (a.name[i] != b.name[i]) return true;
	return false;
}
AGPU_HD void qname_key128(const Rec& r, uint64_t& low, uint64_t& high) {
	uint64_t h = 1469598103934665603ull, g = 0x9AE16A3B2F90404Full;
	for (uint32_t i = 0; i < r.l_read_name && r.name[i]; ++i) { h = (h ^ r.name[i]) * 1099511628211ull; g = (g ^ (r.name[i] + 0x9Eu)) * 0x100000001B3ull; g ^= g >> 29; }
	h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33;
	g ^= g >> 31; g *= 0x7FB5D329728EA185ull; g ^= g >> 27; g *= 0x81DADEF4BC2DD44Dull; g ^= g >> 33;
	low = h; high = g;
}

// 64-bit key of "QNAME,HI": FNV-1a over the name bytes, the hit index mixed in, finalised (murmur3 fmix64); ~0 is reserved for records that take no part
AGPU_HD uint64_t name_key(const Rec& r, int64_t hit_index, uint64_t seed) {
	uint64_t h = 1469598103934665603ull ^ seed;
	for (uint32_t i = 0; i < r.l_read_name && r.name[i]; ++i) h = (h ^ r.name[i]) * 1099511628211ull;
	h ^= (uint64_t) hit_index * 0x9E3779B97F4A7C15ull;
	h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33;
	return h == ~0ull ? ~0ull - 1 : h;
}
AGPU_HD bool same_name(const Rec& a, uint32_t length_a /* qname_length(a) */, int64_t hi_a, const Rec& b, int64_t hi_b) {
	if (hi_a != hi_b) return false;
	if (length_a != qname_length(b)) return false;
	return first_difference(a.name, b.name, length_a) == length_a;
}
AGPU_HD bool same_name(const Rec& a, int64_t hi_a, const Rec& b, int64_t hi_b) { return same_name(a, qname_length(a), hi_a, b, hi_b); }

// ---- plans: a fragment as references into the stream ----------------------------------------------------------------------------------------------

enum : uint8_t { CLIP_NONE = 0, CLIP_START = 1, CLIP_END = 2, PLAN_TANDEM = 3 };
const uint32_t NO_RECORD = 0xFFFFFFFFu;

struct PlanEntry { uint32_t record; uint16_t cigar_index; uint8_t clip; uint8_t supplementary; };
// the alignment is_tandem_duplication synthesises (source/read_chimeric_alignments.cpp:316-334)
struct TandemAlignment { int32_t start, end; uint32_t cigar[3]; uint32_t record; uint8_t n_cigar, strand, first_in_pair, supplementary; };
struct FragmentPlan { // mates_t while it is being filled: the first three alignments and how many were pushed
	PlanEntry entry[3];
	uint8_t count;        // saturates at 255; more than three alignments are malformed anyway
	uint8_t single_end, duplicate, reserved;
};
struct TandemPlan { FragmentPlan plan; TandemAlignment tandem; };

AGPU_HD void plan_clear(FragmentPlan& plan) { plan.count = 0; plan.single_end = 0; plan.duplicate = 0; plan.reserved = 0; for (int k = 0; k < 3; ++k) { plan.entry[k].record = NO_RECORD; plan.entry[k].cigar_index = 0; plan.entry[k].clip = 0; plan.entry[k].supplementary = 0; } }
AGPU_HD void plan_push(FragmentPlan& plan, uint32_t record, const Rec& r, bool is_supplementary, uint32_t cigar_index, uint8_t clip) { // add_chimeric_alignment
	plan.single_end = !(r.flag & BAMF_PAIRED);
	plan.duplicate = plan.duplicate || (r.flag & BAMF_DUP);
	if (plan.count < 3) {
		PlanEntry& e = plan.entry[plan.count];
		e.record = record; e.cigar_index = (uint16_t) cigar_index; e.clip = clip; e.supplementary = is_supplementary;
	}
	if (plan.count < 255) ++plan.count;
}
AGPU_HD void plan_push_tandem(FragmentPlan& plan) {
	if (plan.count < 3) { PlanEntry& e = plan.entry[plan.count]; e.record = NO_RECORD; e.cigar_index = 0; e.clip = PLAN_TANDEM; e.supplementary = 0; }
	if (plan.count < 255) ++plan.count;
}

// an alignment_t derived from a plan entry (source/common.hpp:191-207); its CIGAR is the record's, cut as add_chimeric_alignment cuts it, with up to
// two elements replaced afterwards (remove_malformed_alignments rewrites clipped elements)
struct Aln {
	const uint8_t* cigar_bytes; // the CIGAR of the source record in the stream (round 5: the whole decoded record -- five pointers and six words -- used to sit here, three times per
	                            // fragment, in the scratch memory of the replay and pack kernels; whoever needs more of the record loads it by its index)
	uint32_t record;         // its index, NO_RECORD for a tandem alignment without sequence
	uint32_t sequence_record; // record whose bases are this alignment's `sequence`, NO_RECORD = empty
	int32_t sequence_length;
	int32_t contig, start, end;
	uint32_t n_cigar;
	uint32_t cigar_index;
	uint32_t patch_index[2], patch_value[2];
	uint32_t tandem_cigar[3];
	uint32_t made_index, made_value, shift; // the one element add_chimeric_alignment makes up when it cuts a CIGAR (the clipped rest), and where the others stand in the record's
	uint8_t clip, n_patch;
	bool supplementary, first_in_pair, strand;
	// (worked out once, by materialize: this accessor is inlined at some forty places of the sanity check, and with the loop over the CIGAR that finds the length of the
	// clipped rest inside it the check alone was 49 KB of the replay kernel's instructions)
	AGPU_HD uint32_t base_cigar(uint32_t i) const {
		if (clip == PLAN_TANDEM) return i == 0 ? tandem_cigar[0] : i == 1 ? tandem_cigar[1] : tandem_cigar[2];
		if (i == made_index) return made_value;
		return load_u32(cigar_bytes + 4 * (size_t) (i + shift));
	}
	AGPU_HD uint32_t cigar(uint32_t i) const {
		uint32_t value = base_cigar(i);
		for (uint32_t k = 0; k < n_patch; ++k) if (patch_index[k] == i) value = patch_value[k]; // later rewrites win
		return value;
	}
	AGPU_HD void set_cigar(uint32_t i, uint32_t value) {
		for (uint32_t k = 0; k < n_patch; ++k) if (patch_index[k] == i) { patch_value[k] = value; return; }
		if (n_patch < 2) { patch_index[n_patch] = i; patch_value[n_patch] = value; ++n_patch; }
	}
	AGPU_HD uint32_t preclipping() const { const uint32_t c = cigar(0), op = c & 15; return (op == CIGAR_S || op == CIGAR_H) ? c >> 4 : 0; }
	AGPU_HD uint32_t postclipping() const { const uint32_t c = cigar(n_cigar - 1), op = c & 15; return (op == CIGAR_S || op == CIGAR_H) ? c >> 4 : 0; }
};

AGPU_HD Aln materialize(const IngestStream& in, const PlanEntry& e, const TandemAlignment* tandem) {
	Aln a;
	a.n_patch = 0; a.patch_index[0] = a.patch_index[1] = 0; a.patch_value[0] = a.patch_value[1] = 0;
	a.tandem_cigar[0] = a.tandem_cigar[1] = a.tandem_cigar[2] = 0;
	a.clip = e.clip; a.cigar_index = e.cigar_index;
	a.made_index = NO_RECORD; a.made_value = 0; a.shift = 0;
	if (e.clip == PLAN_TANDEM) {
		const Rec source = load_record(in, tandem->record);
		a.cigar_bytes = source.cigar_bytes;
		a.record = NO_RECORD;
		a.supplementary = tandem->supplementary; a.first_in_pair = tandem->first_in_pair; a.strand = tandem->strand;
		a.contig = source.contig; a.start = tandem->start; a.end = tandem->end;
		a.n_cigar = tandem->n_cigar;
		for (int k = 0; k < 3; ++k) a.tandem_cigar[k] = tandem->cigar[k];
		a.sequence_record = a.supplementary ? NO_RECORD : tandem->record;
		a.sequence_length = a.supplementary ? 0 : source.l_seq;
		return a;
	}
	const Rec r = load_record(in, e.record);
	a.cigar_bytes = r.cigar_bytes;
	a.record = e.record;
	a.strand = r.forward(); a.first_in_pair = r.flag & BAMF_READ1; a.contig = r.contig; a.supplementary = e.supplementary;
	a.sequence_record = e.supplementary ? NO_RECORD : e.record;
	a.sequence_length = e.supplementary ? 0 : r.l_seq;
	if (e.clip == CLIP_START) {
		a.start = r.pos + r.rlen(e.cigar_index); a.end = r.endpos() - 1;
		a.n_cigar = r.n_cigar - e.cigar_index + 1;
		a.made_index = 0; a.made_value = (uint32_t) r.qlen(e.cigar_index) << 4 | (r.op(0) == CIGAR_H ? CIGAR_H : CIGAR_S); // element i >= 1 is the record's cigar_index + i - 1
		a.shift = (uint32_t) e.cigar_index - 1u;
	} else if (e.clip == CLIP_END) {
		a.start = r.pos; a.end = r.pos + r.rlen((uint32_t) e.cigar_index + 1) - 1;
		a.n_cigar = (uint32_t) e.cigar_index + 2;
		a.made_index = (uint32_t) e.cigar_index + 1; a.made_value = (uint32_t) (r.l_seq - r.qlen((uint32_t) e.cigar_index + 1)) << 4 | (r.op(r.n_cigar - 1) == CIGAR_H ? CIGAR_H : CIGAR_S);
	} else {
		a.start = r.pos; a.end = r.endpos() - 1;
		a.n_cigar = r.n_cigar;
	}
	return a;
}

// ---- the helpers of the loop body ------------------------------------------------------------------------------------------------------------------

// reference: source/read_chimeric_alignments.cpp:511-522
AGPU_HD bool is_clipped_at_correct_end(const Rec& r) {
	if (!(r.flag & BAMF_PAIRED)) return true;
	if (r.n_cigar == 0) return false;
	uint32_t clipped_end;
	if (r.flag & BAMF_SUPPLEMENTARY) clipped_end = r.forward() ? r.n_cigar - 1 : 0;
	else clipped_end = r.forward() ? 0 : r.n_cigar - 1;
	const uint32_t op = r.op(clipped_end);
	return op == CIGAR_S || op == CIGAR_H;
}

// reference: source/read_chimeric_alignments.cpp:526-558 (the comparison of characters is a comparison of 4-bit codes)
AGPU_HD bool is_pristine_alignment(const Rec& r) {
	for (uint32_t i = 0; i < r.n_cigar; ++i) {
		const uint32_t op = r.op(i);
		if (op != CIGAR_N && op != CIGAR_M && op != CIGAR_X) return false;
	}
	const uint32_t size = (uint32_t) r.l_seq;
	for (uint32_t i = 2, repeat = 0, count = 1; i + 2 < size; i += 2) {
		if (r.code(i) == r.code(repeat) && r.code(i + 1) == r.code(repeat + 1)) {
			count++;
		} else if (r.code(i + 1) == r.code(repeat + 1) && r.code(i + 2) == r.code(repeat + 2)) {
			count++;
			i++;
		} else {
			count = 1;
			repeat = i;
		}
		if (count >= 8) return false;
	}
	return true;
}

// reference: source/read_chimeric_alignments.cpp:197-211
AGPU_HD bool clipped_sequence_is_adapter(const Rec& mate1, const Rec* mate2) {
	if (mate2 == nullptr) return false;
	if (mate1.pos == mate2->pos && mate1.n_cigar > 0 && mate2->n_cigar > 0) {
		if (!mate1.forward() && mate1.op(0) == CIGAR_S && mate2->forward() && mate2->op(mate2->n_cigar - 1) == CIGAR_S && mate1.len(0) == mate2->len(mate2->n_cigar - 1)) return true;
		if (!mate2->forward() && mate2->op(0) == CIGAR_S && mate1.forward() && mate1.op(mate1.n_cigar - 1) == CIGAR_S && mate2->len(0) == mate1.len(mate1.n_cigar - 1)) return true;
	}
	return false;
}

// "=ACMGRSVTWYHKDBN"[code] without a table in memory: the sixteen characters in two words
AGPU_HD char base_character(uint32_t code) {
	const uint64_t low = 0x565352474D43413Dull /* "=ACMGRSV" */, high = 0x4E42444B48595754ull /* "TWYHKDBN" */;
	return (char) (((code & 8) ? high : low) >> (8 * (code & 7)));
}
// one bit per byte of x that is not zero (bit i = byte i, the lowest address first)
AGPU_HD uint32_t nonzero_bytes(uint64_t x) {
	const uint64_t high = (x | ((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full)) & 0x8080808080808080ull;
	return (uint32_t) (((high >> 7) * 0x0102040810204080ull) >> 56);
}

// reference: source/read_chimeric_alignments.cpp:215-336 (integer types as there: its mixed signed/unsigned arithmetic is part of the result)
AGPU_HD bool is_tandem_duplication(const Rec* r, uint32_t record, const GenomeView& genome, const uint32_t max_itd_length, TandemAlignment& tandem) {
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

	// The window is ~90 positions, and at each the loop below compares bases until the second mismatch behind the sixth base: ~9 bases, two loads each (a character of the
	// contig, a nibble of the read) -- 1 600 loads per clipped read, most of what the whole replay kernel loads.  The first sixteen bases of the clipped segment in the order of
	// that loop are therefore kept in two words, the sixteen characters of the contig they would meet are two loads per position, and where the second counted mismatch lies among
	// them (all but none of the positions) the outcome of the loop is known: it breaks there with fewer than min_alignment_length matches, so the position is a hit only if
	// matches + 2 == clipped_length -- if that cannot be, on to the next position; everything else goes through the loop as before.
	const unsigned int held = clipped_length < 16 ? clipped_length : 16;
	uint64_t read_low = 0, read_high = 0;
	for (unsigned int i = 0; i < held; ++i) {
		const int read_pos = (direction == +1) ? (int) i : (int) (clipped_length - 1 - i);
		const uint64_t character = (uint8_t) base_character(r->code((int32_t) clipped_position + read_pos));
		if (i < 8) read_low |= character << (8 * i); else read_high |= character << (8 * (i - 8));
	}
	const uint32_t held_bits = (1u << held) - 1;
	for (int contig_pos = window_start; contig_pos <= window_end; ++contig_pos) {
		if (max_mismatches == 1 && max_non_template_bases == 6) {
			uint64_t contig_low, contig_high; // the characters of the contig that scan positions 0-7 and 8-15 meet
			if (direction == +1) { contig_low = load_u64((const uint8_t*) contig_sequence + contig_pos); contig_high = load_u64((const uint8_t*) contig_sequence + contig_pos + 8); }
			else {
				const uint8_t* last = (const uint8_t*) contig_sequence + contig_pos + (int) clipped_length - 1;
				contig_low = __builtin_bswap64(load_u64(last - 7)); contig_high = __builtin_bswap64(load_u64(last - 15));
			}
			const uint32_t differs = (nonzero_bytes(contig_low ^ read_low) | nonzero_bytes(contig_high ^ read_high) << 8) & held_bits;
			const uint32_t counted = differs & ~0x3Fu, second = counted & (counted - 1);
			if (second != 0) { // the loop would break at the second counted mismatch
				const uint32_t at = (uint32_t) __builtin_ctz(second);
				if ((uint32_t) __builtin_popcount(~differs & ((1u << at) - 1)) + 2 != clipped_length) continue;
			}
		}
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

// reference: source/read_chimeric_alignments.cpp:19-41
AGPU_HD bool find_spanning_intron(const Rec& r, int32_t gene1_end, int32_t gene2_start, uint32_t& cigar_index, int32_t& read_pos) {
	if (r.n_cigar < 3) return false;
	int32_t before = r.pos, after;
	for (uint32_t i = 0; i < r.n_cigar; ++i) {
		const uint32_t op = r.op(i);
		const uint32_t op_length = op_consumes_reference(op) ? r.len(i) : 0;
		after = before + (int32_t) op_length;
		if (op == CIGAR_N && ((before <= gene1_end && after > gene1_end) || (before < gene2_start && after >= gene2_start))) {
			cigar_index = i;
			read_pos = r.qlen(i);
			return true;
		}
		before = after;
	}
	return false;
}

// genes at one position (get_annotation_by_coordinate with start == end: one bucket of the index, ascending ids)
AGPU_HD ListRef genes_at(const FlatIndexView& gene_index, int32_t contig, int32_t position) {
	if (contig < 0 || (uint32_t) contig >= gene_index.n_contigs) return empty_list();
	const uint32_t k = index_lower_bound(gene_index, (uint32_t) contig, position);
	if (k == gene_index.contig_offset[contig + 1]) return empty_list();
	return index_bucket(gene_index, k);
}
AGPU_HD bool lists_intersect(ListRef a, ListRef b) {
	uint32_t i = 0, j = 0;
	while (i < a.n && j < b.n) { const uint32_t x = a.p[i], y = b.p[j]; if (x < y) ++i; else if (y < x) ++j; else return true; }
	return false;
}
// reference: get_boundaries_of_biggest_gene, source/annotation.cpp:558-567
AGPU_HD void boundaries_of_biggest_gene(const AnnotationView& ann, ListRef genes, int32_t& start, int32_t& end) {
	start = -1; end = -1;
	for (uint32_t g = 0; g < genes.n; ++g) {
		const int32_t gene_start = ann.gene_start[genes.p[g]], gene_end = ann.gene_end[genes.p[g]];
		if (start == -1 || start > gene_start) start = gene_start;
		if (end == -1 || end < gene_end) end = gene_end;
	}
}

// reference: source/read_chimeric_alignments.cpp:93-193.  `record` is never null; `previous` may be.
AGPU_HD bool extract_read_through_alignment(FragmentPlan& plan, uint32_t record_index, const Rec& record, uint32_t previous_index, const Rec* previous, const AnnotationView& ann) {
	const Rec* forward_mate = &record; const Rec* reverse_mate = previous;
	uint32_t forward_index = record_index, reverse_index = previous_index;
	if (!forward_mate->forward()) { const Rec* t = forward_mate; forward_mate = reverse_mate; reverse_mate = t; const uint32_t ti = forward_index; forward_index = reverse_index; reverse_index = ti; }
	ListRef forward_genes, reverse_genes;
	if (forward_mate != nullptr) forward_genes = genes_at(ann.gene_index, forward_mate->contig, forward_mate->pos);
	else forward_genes = genes_at(ann.gene_index, reverse_mate->contig, reverse_mate->pos);
	if (reverse_mate != nullptr) reverse_genes = genes_at(ann.gene_index, reverse_mate->contig, reverse_mate->endpos());
	else reverse_genes = genes_at(ann.gene_index, forward_mate->contig, forward_mate->endpos());
	const bool common_empty = !lists_intersect(forward_genes, reverse_genes);
	if (!(common_empty && !(forward_genes.n == 0 && reverse_genes.n == 0))) return false;

	int32_t forward_gene_start, forward_gene_end, reverse_gene_start, reverse_gene_end;
	boundaries_of_biggest_gene(ann, forward_genes, forward_gene_start, forward_gene_end);
	boundaries_of_biggest_gene(ann, reverse_genes, reverse_gene_start, reverse_gene_end);
	if (forward_gene_end == -1) forward_gene_end = reverse_gene_start - 1;
	if (reverse_gene_start == -1) reverse_gene_start = forward_gene_end + 1;

	uint32_t forward_cigar_op = 0, reverse_cigar_op = 0;
	int32_t forward_read_pos = 0, reverse_read_pos = 0;
	const bool forward_has_intron = (forward_mate == nullptr) ? false : find_spanning_intron(*forward_mate, forward_gene_end, reverse_gene_start, forward_cigar_op, forward_read_pos);
	const bool reverse_has_intron = (reverse_mate == nullptr) ? false : find_spanning_intron(*reverse_mate, forward_gene_end, reverse_gene_start, reverse_cigar_op, reverse_read_pos);
	const bool is_new = plan.count == 0; // fragments.insert(...).second
	if (forward_has_intron && (!reverse_has_intron || forward_read_pos < reverse_mate->l_seq - reverse_read_pos)) {
		if (is_new) {
			plan_push(plan, forward_index, *forward_mate, false, forward_cigar_op + 1, CLIP_START);
			plan_push(plan, forward_index, *forward_mate, true, forward_cigar_op - 1, CLIP_END);
			if (reverse_mate != nullptr) {
				if (reverse_has_intron) plan_push(plan, reverse_index, *reverse_mate, false, reverse_cigar_op + 1, CLIP_START);
				else plan_push(plan, reverse_index, *reverse_mate, false, 0, CLIP_NONE);
			}
			return true;
		}
	} else if (reverse_has_intron) {
		if (is_new) {
			plan_push(plan, reverse_index, *reverse_mate, true, reverse_cigar_op + 1, CLIP_START);
			plan_push(plan, reverse_index, *reverse_mate, false, reverse_cigar_op - 1, CLIP_END);
			if (forward_mate != nullptr) {
				if (forward_has_intron) plan_push(plan, forward_index, *forward_mate, false, forward_cigar_op - 1, CLIP_END);
				else plan_push(plan, forward_index, *forward_mate, false, 0, CLIP_NONE);
			}
			return true;
		}
	} else if (forward_mate != nullptr && reverse_mate != nullptr && reverse_mate->pos >= reverse_gene_start && forward_mate->endpos() <= forward_gene_end) {
		if (is_new) {
			plan_push(plan, forward_index, *forward_mate, false, 0, CLIP_NONE);
			plan_push(plan, reverse_index, *reverse_mate, false, 0, CLIP_NONE);
		}
		return true;
	}
	return false;
}

// ---- coverage (reference: coverage_t::add_fragment, source/read_stats.cpp:161-266) ----------------------------------------------------------------
// The windows are shared by all threads: +1 with saturation commutes, so they are counted in 32 bits and clamped to the reference's 16 bits when the ingest is over; the
// start/end flags are plain stores of 1.  The windows an aligned block covers are a RANGE [lo, hi): instead of one atomic per window (eight for a read of 150 bases -- every
// one of them a 64-byte request to the fabric, 2 x 690 GB at 10^8 fragments, profiles/r04y_pmc100m_pmc_summary.txt) the range is noted in a difference array, +1 at lo and
// -1 at hi, neighbouring ranges of a fragment (its blocks behind an insertion, its overlapping mates) as one; a prefix sum behind the last record turns the differences into
// the counts (coverage_from_differences_kernel).  A contig has one slot more than windows in that array -- the -1 behind its last window -- so window w of contig c is slot
// window_offset[c] + c + w, every contig sums to zero and one prefix sum over the whole array serves all contigs.

struct CoverageBuild {
	uint32_t n_contigs;
	const uint64_t* window_offset; // [n_contigs + 1]; a contig without sequence has no windows
	uint32_t* windows;             // the difference array: window_offset[n_contigs] + n_contigs slots
	uint8_t* fragment_starts;
	uint8_t* fragment_ends;
};
AGPU_HD uint64_t coverage_difference_slots(uint64_t windows, uint32_t n_contigs) { return windows + n_contigs; }

AGPU_HD void coverage_add(uint32_t* slot, uint32_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
	__hip_atomic_fetch_add(slot, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
	__atomic_fetch_add(slot, value, __ATOMIC_RELAXED);
#endif
}
// a range of windows waiting for its neighbour: [lo, hi) behind slot `base` of the difference array
struct CoverageRange {
	uint64_t base; int64_t lo, hi;
	AGPU_HD void flush(uint32_t* differences) { if (hi > lo) { coverage_add(&differences[base + (uint64_t) lo], 1u); coverage_add(&differences[base + (uint64_t) hi], 0xFFFFFFFFu); } hi = lo = 0; }
	AGPU_HD void add(uint32_t* differences, uint64_t to_base, int64_t from, int64_t to) { // windows [from, to), to > from
		if (hi > lo && to_base == base && from == hi) { hi = to; return; }
		flush(differences);
		base = to_base; lo = from; hi = to;
	}
};

AGPU_HD void add_fragment_to_coverage(const CoverageBuild& coverage, const Rec& mate1, uint16_t flag1, const Rec* mate2_or_null, bool is_chimeric) {
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
	const uint64_t slots1 = begin1 + (uint64_t) mate1.contig, slots2 = begin2 + (uint64_t) mate2.contig; // (one slot more per contig in front of this one)
	CoverageRange pending; pending.base = 0; pending.lo = pending.hi = 0;
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
			op = op1; begin = slots1; size = size1; position1 += (int32_t) length1; position = position1;
		} else if (i2 < mate2.n_cigar) {
			i2++;
			if (length2 == 0) continue;
			op = op2; begin = slots2; size = size2; position2 += (int32_t) length2; position = position2;
		} else {
			break;
		}
		if (op_consumes_query(op & 15)) {
			// the reference's loop -- while (window <= position / resolution) { if (window inside the contig && position - window * resolution >= resolution / 2) ++coverage[window]; ++window; } --
			// in closed form: the condition holds for every window below the last one and for the last one if the block reaches its middle; nothing is counted at a negative position
			const int32_t last = position / COVERAGE_RESOLUTION;
			if (window <= last) {
				if (position >= 0) {
					const int64_t from = window > 0 ? window : 0;
					int64_t to = (position - last * COVERAGE_RESOLUTION >= COVERAGE_RESOLUTION / 2) ? (int64_t) last + 1 : (int64_t) last;
					if (to > (int64_t) size) to = (int64_t) size;
					if (to > from) pending.add(coverage.windows, begin, from, to);
				}
				window = last + 1;
			}
		} else {
			window = position / COVERAGE_RESOLUTION;
		}
	}
	pending.flush(coverage.windows);
	if (!is_chimeric) {
		if ((flag1 & BAMF_REVERSE) || !(flag1 & BAMF_PAIRED)) { const uint64_t w = (uint64_t) ((position1 - 1) / COVERAGE_RESOLUTION); if (position1 >= 1 && w < size1) coverage.fragment_ends[begin1 + w] = 1; }
		else { const uint64_t w = (uint64_t) ((position2 - 1) / COVERAGE_RESOLUTION); if (position2 >= 1 && w < size2) coverage.fragment_ends[begin2 + w] = 1; }
	}
}

// ---- the loop body over the records of one read name ----------------------------------------------------------------------------------------------

struct IngestContext {
	IngestStream stream;
	AnnotationView annotation;   // GTF genes (gene_index before the dummy genes)
	GenomeView genome;           // contig_bits: interesting / viral
	CoverageBuild coverage;
	const uint8_t* record_bits;  // RECORD_* per record
	uint32_t max_itd_length;
	uint8_t external_duplicate_marking;
};

struct GroupTally { uint32_t malformed; uint32_t chimeric; uint32_t collision; }; // malformed_count, !no_chimeric_reads; a record whose name is not the one of the first record of its group (two names, one key)

// `records` = the indices of the active records of one "QNAME,HI" in stream order.  plain = fragments[read_name], itd = fragments[read_name + "ITD"].
// viral_reads[contig] += pristine reads (64-bit counters).
// names_of != nullptr: the records were put together by their keys and nobody has compared their names yet -- every record against *names_of, the first one, while it is at hand
template <class ViralCounter> AGPU_HD void replay_group(const IngestContext& ctx, const uint32_t* records, uint32_t n_records, FragmentPlan& plain, TandemPlan& itd, GroupTally& tally, ViralCounter& count_viral_read,
                                                        const Rec* names_of = nullptr, uint32_t length_of_names = 0 /* qname_length(*names_of) */, int64_t hit_index_of_names = 0) {
	plan_clear(plain); plan_clear(itd.plan);
	itd.tandem.start = 0; itd.tandem.end = 0; itd.tandem.cigar[0] = itd.tandem.cigar[1] = itd.tandem.cigar[2] = 0; itd.tandem.record = NO_RECORD;
	itd.tandem.n_cigar = 0; itd.tandem.strand = 0; itd.tandem.first_in_pair = 0; itd.tandem.supplementary = 0;
	uint32_t parked = NO_RECORD; // the first mate waiting for the second (the reference's collated_bam_records)
	for (uint32_t k = 0; k < n_records; ++k) {
		const uint32_t index = records[k];
		const Rec record = load_record(ctx.stream, index);
		if (names_of != nullptr && k > 0 && !same_name(*names_of, length_of_names, hit_index_of_names, record, hit_index_of(ctx.stream, index, record))) tally.collision = 1;
		if (record.flag & BAMF_SUPPLEMENTARY) {
			if (is_clipped_at_correct_end(record)) plan_push(plain, index, record, true, 0, CLIP_NONE);
			else tally.malformed++;
			tally.chimeric = 1;
			continue;
		}
		// (every function of the loop body is called from one place: each call site is a copy of the function in the kernel, which is 90 KB of instructions as it is)
		uint32_t previous_index = NO_RECORD;
		Rec previous_storage;
		const Rec* previous = nullptr;
		bool is_read_through = false;
		uint16_t coverage_flag = record.flag;
		const bool discordant_mate = (record.flag & BAMF_PAIRED) && !(record.flag & BAMF_PROPER_PAIR);
		if (discordant_mate) {
			plan_push(plain, index, record, false, 0, CLIP_NONE);
			tally.chimeric = 1;
			coverage_flag = 0; is_read_through = true; // (the reference zeroes the flag word and passes is_chimeric = true)
		} else {
			if (record.flag & BAMF_PAIRED) {
				if (parked == NO_RECORD) { parked = index; continue; }
				previous_index = parked; parked = NO_RECORD;
				previous_storage = load_record(ctx.stream, previous_index);
				previous = &previous_storage;
			}
			bool is_tandem_alignment = false;
			if (!clipped_sequence_is_adapter(record, previous) && (previous == nullptr || record.forward() != previous->forward())) {
				TandemAlignment tandem;
				bool found = false;
				AGPU_NOUNROLL for (uint32_t mate = 0; mate < 2 && !found; ++mate) found = is_tandem_duplication(mate ? previous : &record, mate ? previous_index : index, ctx.genome, ctx.max_itd_length, tandem);
				if (found) {
					plan_push(itd.plan, index, record, record.forward() == (bool) tandem.strand && !tandem.supplementary, 0, CLIP_NONE);
					if (previous != nullptr) plan_push(itd.plan, previous_index, *previous, previous->forward() == (bool) tandem.strand && !tandem.supplementary, 0, CLIP_NONE);
					if (itd.plan.count < 3) itd.tandem = tandem; // (a fourth alignment makes the fragment malformed whatever it is)
					plan_push_tandem(itd.plan);
					is_tandem_alignment = true;
				}
			}
			const bool record_has_sa = ctx.record_bits[index] & RECORD_HAS_SA, previous_has_sa = previous != nullptr && (ctx.record_bits[previous_index] & RECORD_HAS_SA);
			if ((record_has_sa && is_clipped_at_correct_end(record)) || (previous_has_sa && is_clipped_at_correct_end(*previous))) {
				plan_push(plain, index, record, false, 0, CLIP_NONE);
				if (previous != nullptr) plan_push(plain, previous_index, *previous, false, 0, CLIP_NONE);
				tally.chimeric = 1;
			} else if (!is_tandem_alignment) {
				is_read_through = extract_read_through_alignment(plain, index, record, previous_index, previous, ctx.annotation);
				if (record.contig >= 0 && (ctx.genome.contig_bits[record.contig] & CBIT_VIRAL)) {
					AGPU_NOUNROLL for (uint32_t mate = 0; mate < 2; ++mate) { const Rec* read = mate ? previous : &record; if (read != nullptr && is_pristine_alignment(*read)) count_viral_read((uint32_t) read->contig); }
				}
			}
		}
		if (!ctx.external_duplicate_marking || !(record.flag & BAMF_DUP)) add_fragment_to_coverage(ctx.coverage, record, coverage_flag, previous, is_read_through);
	}
}

// ---- sanity check and slot order (reference: remove_malformed_alignments, source/read_chimeric_alignments.cpp:377-506) -----------------------------

struct Fragment3 { Aln a[3]; uint32_t n; bool single_end, duplicate; };

// reference: source/read_chimeric_alignments.cpp:340-373
AGPU_HD bool disjoin_split_read_segments(Aln& split_read, Aln& supplementary) {
	const int min_remaining_supplementary_segment = 10;
	const unsigned int clipped_split_read = split_read.strand ? split_read.preclipping() : split_read.postclipping();
	const unsigned int clipped_supplementary = supplementary.strand ? supplementary.postclipping() : supplementary.preclipping();
	const int overlap = (int) split_read.sequence_length - clipped_split_read - clipped_supplementary;
	if (overlap <= 0) return true;
	const unsigned int clipped_op = supplementary.strand ? supplementary.n_cigar - 1 : 0;
	const unsigned int matching_op = supplementary.strand ? clipped_op - 1 : 1;
	if (supplementary.n_cigar < 2 || (supplementary.cigar(matching_op) & 15) != CIGAR_M || (int) (supplementary.cigar(matching_op) >> 4) < overlap + min_remaining_supplementary_segment) return false;
	const uint32_t clipped_element = supplementary.cigar(clipped_op), matching_element = supplementary.cigar(matching_op);
	supplementary.set_cigar(clipped_op, ((clipped_element >> 4) + overlap) << 4 | (clipped_element & 15));
	supplementary.set_cigar(matching_op, ((matching_element >> 4) - overlap) << 4 | (matching_element & 15));
	if (supplementary.strand) supplementary.end -= overlap; else supplementary.start += overlap;
	return true;
}

AGPU_HD void swap_alignments(Aln& x, Aln& y) { const Aln t = x; x = y; y = t; }
AGPU_HD void take_sequence(Aln& target, const Aln& source) { target.sequence_record = source.sequence_record; target.sequence_length = source.sequence_length; }

// materialises the plan and applies the reference's checks; false = malformed (counted by the caller)
AGPU_HD bool normalize_plan(const IngestStream& in, const FragmentPlan& plan, const TandemAlignment* tandem, Fragment3& f) {
	f.single_end = plan.single_end; f.duplicate = plan.duplicate;
	const uint32_t size = plan.count;
	if (size < 2 || size > 3) return false; // every other size is rejected below in the reference, too (single-end wants 2, paired-end 2 or 3)
	for (uint32_t k = 0; k < size; ++k) f.a[k] = materialize(in, plan.entry[k], tandem);
	for (uint32_t k = 0; k < size; ++k) if (f.a[k].n_cigar == 0) return false; // (the reference reads cigar[0] of an empty CIGAR: undefined there)
	Aln* a = f.a;
	f.n = size;
	if (plan.single_end) {
		if (!(size == 2 && (a[MATE1].supplementary != a[MATE2].supplementary))) return false;
		if (a[MATE1].end - a[MATE1].start > a[MATE2].end - a[MATE2].start) { a[2] = a[MATE2]; a[MATE2] = a[MATE1]; }
		else { a[2] = a[MATE1]; a[MATE1] = a[MATE2]; }
		f.n = 3;
		if (!a[MATE1].supplementary) take_sequence(a[SPLIT_READ], a[MATE1]);
		else if (!a[SPLIT_READ].supplementary) take_sequence(a[MATE1], a[SPLIT_READ]);
		else { take_sequence(a[MATE1], a[SUPPLEMENTARY]); take_sequence(a[SPLIT_READ], a[SUPPLEMENTARY]); }
		a[SUPPLEMENTARY].sequence_record = NO_RECORD; a[SUPPLEMENTARY].sequence_length = 0;
		// hard clips become soft clips
		if ((a[MATE1].cigar(0) & 15) == CIGAR_H) a[MATE1].set_cigar(0, a[MATE1].cigar(0) >> 4 << 4 | CIGAR_S);
		if ((a[MATE1].cigar(a[MATE1].n_cigar - 1) & 15) == CIGAR_H) a[MATE1].set_cigar(a[MATE1].n_cigar - 1, a[MATE1].cigar(a[MATE1].n_cigar - 1) >> 4 << 4 | CIGAR_S);
		if ((a[SPLIT_READ].cigar(0) & 15) == CIGAR_H) a[SPLIT_READ].set_cigar(0, a[SPLIT_READ].cigar(0) >> 4 << 4 | CIGAR_S);
		if ((a[SPLIT_READ].cigar(a[SPLIT_READ].n_cigar - 1) & 15) == CIGAR_H) { // the length is taken from MATE1's CIGAR at SPLIT_READ's last index, as the reference does (:415)
			const uint32_t last = a[SPLIT_READ].n_cigar - 1;
			if (last >= a[MATE1].n_cigar) return false; // (std::vector::at throws there)
			a[SPLIT_READ].set_cigar(last, a[MATE1].cigar(last) >> 4 << 4 | CIGAR_S);
		}
		a[SUPPLEMENTARY].supplementary = true; a[MATE1].supplementary = false; a[SPLIT_READ].supplementary = false;
		bool flip_mate1_strand;
		const bool same = a[SPLIT_READ].strand == a[SUPPLEMENTARY].strand;
		const uint64_t length = (uint64_t) (uint32_t) a[SPLIT_READ].sequence_length; // size_t arithmetic in the reference
		if (length - a[SPLIT_READ].preclipping() - (same ? a[SUPPLEMENTARY].postclipping() : a[SUPPLEMENTARY].preclipping()) <
		    length - a[SPLIT_READ].postclipping() - (same ? a[SUPPLEMENTARY].preclipping() : a[SUPPLEMENTARY].postclipping()))
			flip_mate1_strand = a[SPLIT_READ].strand == true;
		else
			flip_mate1_strand = a[SPLIT_READ].strand == false;
		a[MATE1].strand = complement_strand_if(a[MATE1].strand, flip_mate1_strand);
		a[SPLIT_READ].strand = complement_strand_if(a[SPLIT_READ].strand, !flip_mate1_strand);
		a[SUPPLEMENTARY].strand = complement_strand_if(a[SUPPLEMENTARY].strand, !flip_mate1_strand);
		a[MATE1].first_in_pair = !flip_mate1_strand;
		a[SPLIT_READ].first_in_pair = flip_mate1_strand;
		a[SUPPLEMENTARY].first_in_pair = flip_mate1_strand;
		if (!disjoin_split_read_segments(a[SPLIT_READ], a[SUPPLEMENTARY])) return false;
	} else {
		if (size == 3) {
			if (a[MATE1].supplementary) swap_alignments(a[MATE1], a[SUPPLEMENTARY]);
			else if (a[MATE2].supplementary) swap_alignments(a[MATE2], a[SUPPLEMENTARY]);
			if (a[SPLIT_READ].first_in_pair != a[SUPPLEMENTARY].first_in_pair) swap_alignments(a[MATE1], a[MATE2]);
			if (a[MATE1].supplementary || a[SPLIT_READ].supplementary || !a[SUPPLEMENTARY].supplementary) return false;
			if (a[MATE1].contig != a[SPLIT_READ].contig || a[MATE1].strand == a[SPLIT_READ].strand) return false;
			if (!disjoin_split_read_segments(a[SPLIT_READ], a[SUPPLEMENTARY])) return false;
		} else {
			if (a[MATE1].supplementary || a[MATE2].supplementary) return false;
		}
	}
	if ((a[MATE1].cigar(0) & 15) == CIGAR_H || (a[MATE1].cigar(a[MATE1].n_cigar - 1) & 15) == CIGAR_H ||
	    (a[MATE2].cigar(0) & 15) == CIGAR_H || (a[MATE2].cigar(a[MATE2].n_cigar - 1) & 15) == CIGAR_H)
		return false;
	return true;
}

// ---- names ("QNAME,HI" [+ "ITD"]: the key of the reference's std::map, hazard H3) -------------------------------------------------------------------

AGPU_HD uint32_t decimal_digits(int64_t value, char* out /* [21] */) { // std::to_string(long)
	char reversed[21]; uint32_t n = 0;
	uint64_t magnitude = value < 0 ? (uint64_t) (-(value + 1)) + 1 : (uint64_t) value;
	do { reversed[n++] = (char) ('0' + magnitude % 10); magnitude /= 10; } while (magnitude > 0);
	uint32_t at = 0;
	if (value < 0) out[at++] = '-';
	while (n > 0) out[at++] = reversed[--n];
	return at;
}

struct FragmentName { const uint8_t* qname; uint32_t qname_length; char suffix[28]; uint32_t suffix_length; }; // suffix = "," + HI + ["ITD"]
const uint32_t LENGTH_UNKNOWN = 0xFFFFFFFFu;
AGPU_HD FragmentName fragment_name(const Rec& representative, int64_t hit_index /* hit_index_of(representative) */, bool itd, uint32_t length = LENGTH_UNKNOWN /* qname_length(representative), where the caller has it */) {
	FragmentName name;
	name.qname = representative.name; name.qname_length = length != LENGTH_UNKNOWN ? length : qname_length(representative);
	name.suffix[0] = ',';
	name.suffix_length = 1 + decimal_digits(hit_index, name.suffix + 1);
	if (itd) { name.suffix[name.suffix_length++] = 'I'; name.suffix[name.suffix_length++] = 'T'; name.suffix[name.suffix_length++] = 'D'; }
	return name;
}
AGPU_HD uint32_t name_length(const FragmentName& name) { return name.qname_length + name.suffix_length; }
AGPU_HD uint8_t name_byte(const FragmentName& name, uint32_t i) { return i < name.qname_length ? name.qname[i] : (i - name.qname_length < name.suffix_length ? (uint8_t) name.suffix[i - name.qname_length] : 0); }
// eight bytes of the name from position 8 * chunk, big endian, zero padded: unsigned comparison of the chunks in order == std::string::compare
AGPU_HD uint64_t name_chunk(const FragmentName& name, uint32_t chunk) {
	uint64_t value = 0;
	for (uint32_t k = 0; k < 8; ++k) value = value << 8 | name_byte(name, 8 * chunk + k);
	return value;
}
AGPU_HD int compare_names(const FragmentName& x, const FragmentName& y) {
	const uint32_t nx = name_length(x), ny = name_length(y), n = nx < ny ? nx : ny;
	const uint32_t qnames = x.qname_length < y.qname_length ? x.qname_length : y.qname_length, different = first_difference(x.qname, y.qname, qnames); // as far as both are QNAME: eight bytes at a time
	if (different < qnames) return x.qname[different] < y.qname[different] ? -1 : 1;
	for (uint32_t i = qnames; i < n; ++i) { const uint8_t cx = name_byte(x, i), cy = name_byte(y, i); if (cx != cy) return cx < cy ? -1 : 1; }
	return nx < ny ? -1 : nx > ny ? 1 : 0;
}

// ---- the packed batch: what one fragment writes (reference of the layout: include/arriba_gpu.h: agpu_batch_view) ------------------------------------

struct FragmentSizes { uint32_t cigar_words, sequence_bytes, name_length; };
AGPU_HD uint32_t padded_sequence_bytes(uint32_t bases) { return (((bases + 1) / 2) + 3) & ~3u; } // two bases per byte, every sequence on a 4-byte boundary
AGPU_HD void fragment_sizes(const Fragment3& f, const Rec& representative, int64_t hit_index, bool itd, FragmentSizes& sizes, uint32_t length = LENGTH_UNKNOWN /* qname_length(representative) */) {
	uint32_t cigar_words = 0, sequence_bytes = 0;
	for (uint32_t s = 0; s < f.n; ++s) {
		cigar_words += f.a[s].n_cigar;
		if (s < 2) sequence_bytes += padded_sequence_bytes((uint32_t) f.a[s].sequence_length);
	}
	sizes.cigar_words = cigar_words; sizes.sequence_bytes = sequence_bytes;
	sizes.name_length = name_length(fragment_name(representative, hit_index, itd, length));
}

struct PackTarget {
	uint8_t* n_aln; uint8_t* fbits; uint32_t* group;
	uint16_t* contig[3]; int32_t* start[3]; int32_t* end[3]; uint8_t* abits[3]; uint32_t* cigar_offset[3]; uint16_t* cigar_count[3];
	uint32_t* cigar_pool; uint32_t* seq_offset[2]; uint32_t* seq_length[2]; uint8_t* seq_pool; char* names;
	uint64_t* name_offset;     // of a batch: 64-bit (10^8 fragments with the 45 characters of an Illumina read name are 4.7 GB of names)
	uint32_t* row_name_offset; // of rows gathered for the host (name_offset == nullptr): their names are a small pool of their own
};

// row i of the batch from a sanity-checked fragment; returns the length of its longest read
AGPU_HD uint32_t write_fragment(const IngestStream& in, const Fragment3& f, const FragmentName& name, uint64_t i, uint64_t cigar_at, uint64_t sequence_at, uint64_t name_at, uint32_t group, const PackTarget& out) {
	out.n_aln[i] = (uint8_t) f.n;
	out.fbits[i] = (f.single_end ? FBIT_SINGLE_END : 0) | (f.duplicate ? FBIT_DUPLICATE : 0);
	out.group[i] = group;
	uint32_t longest = 0;
	for (uint32_t s = 0; s < 3; ++s) {
		if (s >= f.n) {
			out.contig[s][i] = 0; out.start[s][i] = 0; out.end[s][i] = 0; out.abits[s][i] = 0; out.cigar_offset[s][i] = 0; out.cigar_count[s][i] = 0;
			if (s < 2) { out.seq_offset[s][i] = 0; out.seq_length[s][i] = 0; }
			continue;
		}
		const Aln& a = f.a[s];
		out.contig[s][i] = (uint16_t) a.contig; out.start[s][i] = a.start; out.end[s][i] = a.end;
		out.abits[s][i] = (a.strand ? ABIT_STRAND : 0) | (a.first_in_pair ? ABIT_FIRST_IN_PAIR : 0) | (a.supplementary ? ABIT_SUPPLEMENTARY : 0) | ABIT_PREDICTED_STRAND_AMBIGUOUS;
		out.cigar_offset[s][i] = (uint32_t) cigar_at; out.cigar_count[s][i] = (uint16_t) a.n_cigar;
		for (uint32_t k = 0; k < a.n_cigar; ++k) out.cigar_pool[cigar_at + k] = a.cigar(k);
		cigar_at += a.n_cigar;
		if (s < 2) {
			const uint32_t length = (uint32_t) a.sequence_length, bytes = (length + 1) / 2, padded = padded_sequence_bytes(length);
			out.seq_offset[s][i] = (uint32_t) (sequence_at / 4); out.seq_length[s][i] = length;
			if (length > longest) longest = length;
			if (length > 0) {
				const Rec source = load_record(in, a.sequence_record);
				uint32_t* target = (uint32_t*) (out.seq_pool + sequence_at); // sequences start on 4-byte boundaries
				for (uint32_t w = 0; w < padded / 4; ++w) {
					uint32_t value = 0;
					if (4 * w + 4 < bytes) value = load_u32(source.seq_bytes + 4 * w);
					else for (uint32_t b = 0; b < 4; ++b) {
						const uint32_t at = 4 * w + b;
						if (at >= bytes) break;
						uint32_t byte = source.seq_bytes[at];
						if (at == bytes - 1 && (length & 1)) byte &= 0xF0u; // the unused low nibble of an odd length
						value |= byte << (8 * b);
					}
					target[w] = value;
				}
			}
			sequence_at += padded;
		}
	}
	out.name_offset[i] = name_at;
	const uint32_t length = name_length(name);
	uint32_t copied = 0; // the QNAME eight bytes at a time (the neighbours of the name in the pool belong to other fragments: only whole words inside it)
	for (; copied + 8 <= name.qname_length; copied += 8) { const uint64_t word = load_u64(name.qname + copied); __builtin_memcpy(out.names + name_at + copied, &word, 8); }
	for (uint32_t k = copied; k < name.qname_length; ++k) out.names[name_at + k] = (char) name.qname[k];
	for (uint32_t k = name.qname_length; k < length; ++k) out.names[name_at + k] = (char) name_byte(name, k);
	return longest;
}

// rows of a resident batch copied out for the host (agpu_gather_rows_*): sizes of row i, then the copy into row k of the target
AGPU_HD void row_sizes(const BatchView& b, const uint64_t* name_offset, uint64_t i, uint32_t& cigar_words, uint32_t& sequence_bytes, uint32_t& name_bytes) {
	cigar_words = 0; sequence_bytes = 0;
	for (uint32_t s = 0; s < b.n_aln[i]; ++s) {
		cigar_words += b.cigar_count[s][i];
		if (s < 2) sequence_bytes += padded_sequence_bytes(b.seq_length[s][i]);
	}
	name_bytes = name_offset ? (uint32_t) (name_offset[i + 1] - name_offset[i]) : 0;
}
AGPU_HD void copy_row(const BatchView& b, const uint8_t* pristine_fbits, const uint8_t* const* pristine_abits, const uint64_t* name_offset, const char* names, uint64_t i, uint64_t k,
                      uint64_t cigar_at, uint64_t sequence_at, uint64_t name_at, const PackTarget& out) {
	const uint32_t n_aln = b.n_aln[i];
	out.n_aln[k] = (uint8_t) n_aln; out.fbits[k] = pristine_fbits[i]; out.group[k] = b.group[i];
	for (uint32_t s = 0; s < 3; ++s) {
		out.contig[s][k] = b.contig[s][i]; out.start[s][k] = b.start[s][i]; out.end[s][k] = b.end[s][i]; out.abits[s][k] = pristine_abits[s][i];
		const uint32_t count = s < n_aln ? b.cigar_count[s][i] : 0;
		out.cigar_count[s][k] = (uint16_t) count; out.cigar_offset[s][k] = s < n_aln ? (uint32_t) cigar_at : 0;
		for (uint32_t c = 0; c < count; ++c) out.cigar_pool[cigar_at + c] = b.cigar_pool[b.cigar_offset[s][i] + c];
		cigar_at += count;
		if (s < 2) {
			const uint32_t length = s < n_aln ? b.seq_length[s][i] : 0, padded = padded_sequence_bytes(length);
			out.seq_length[s][k] = length; out.seq_offset[s][k] = s < n_aln ? (uint32_t) (sequence_at / 4) : 0;
			if (length > 0) {
				const uint32_t* source = (const uint32_t*) b.seq_pool + b.seq_offset[s][i];
				uint32_t* target = (uint32_t*) (out.seq_pool + sequence_at);
				for (uint32_t w = 0; w < padded / 4; ++w) target[w] = source[w];
			}
			sequence_at += padded;
		}
	}
	if (out.name_offset) out.name_offset[k] = name_at; else out.row_name_offset[k] = (uint32_t) name_at;
	if (name_offset) for (uint64_t c = name_offset[i]; c < name_offset[i + 1]; ++c) out.names[name_at + (c - name_offset[i])] = names[c];
}

// detect_strandedness (source/read_stats.cpp:94-143) asks every split read: 0 = not informative, 1 = informative, 3 = informative and on the gene's strand.
// `b.abits`: the strands as ingested; `ann`: the GTF genes.
AGPU_HD uint8_t strandedness_vote(const BatchView& b, const AnnotationView& ann, uint64_t i) {
	if (b.n_aln[i] != 3) return 0;
	const bool split_strand = b.abits[SPLIT_READ][i] & ABIT_STRAND, supplementary_strand = b.abits[SUPPLEMENTARY][i] & ABIT_STRAND;
	int32_t distance = b.start[SPLIT_READ][i] - b.start[SUPPLEMENTARY][i]; if (distance < 0) distance = -distance;
	if (!(b.contig[SPLIT_READ][i] == b.contig[SUPPLEMENTARY][i] && split_strand == supplementary_strand && distance < 400000)) return 0;
	AGPU_IDSET(genes);
	query_by_coordinate(ann.gene_index, b.contig[SPLIT_READ][i], b.start[SPLIT_READ][i], b.end[SPLIT_READ][i], IdentityMap(), genes);
	if (genes.n != 1 || genes.overflow) return 0;
	const uint32_t gene = genes.get(0);
	const int32_t position = split_strand ? b.start[SPLIT_READ][i] : b.end[SPLIT_READ][i];
	if (!is_breakpoint_spliced(ann, gene, split_strand, position)) return 0;
	const bool gene_strand = ann.gene_bits[gene] & GBIT_STRAND, mate1_strand = b.abits[MATE1][i] & ABIT_STRAND;
	const bool matching = ((b.abits[SPLIT_READ][i] & ABIT_FIRST_IN_PAIR) && split_strand == gene_strand) || ((b.abits[MATE1][i] & ABIT_FIRST_IN_PAIR) && mate1_strand == gene_strand);
	return matching ? 3 : 1;
}

}

#endif
