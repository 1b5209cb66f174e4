// shard_host.hpp -- wire format of one part of a sample (agpu_shard_export -> one all-gather -> agpu_shard_merge; include/arriba_gpu.h).
//
// One sample over the GPUs of a node (SURVEY.md section 8 row e, BASELINE.json config 4): every rank runs read_chimeric_alignments
// (source/read_chimeric_alignments.cpp:560-773) over its part of the alignment records; what the reference's loop leaves behind -- chimeric_alignments,
// coverage, mapped_reads, mapped_viral_reads_by_contig, the malformed / missing-HI counters -- is additive over parts that do not share a read name,
// so the parts are put together again by concatenation (fragments, in name order), sums (counters, coverage before its 16-bit saturation) and ORs (the
// start / end flags of coverage_t).  A part travels as ONE block of bytes -- a header of 64-bit fields and the columns of the batch in a fixed order,
// each on a 16-byte boundary -- so that the exchange is a single large collective.
#ifndef AGPU_SHARD_HOST_HPP
#define AGPU_SHARD_HOST_HPP

#include <cstdint>
#include <cstring>

namespace agpu {

const uint64_t SHARD_MAGIC = 0x3144524148534741ull; // "AGSHARD1"

struct ShardHeader {
	uint64_t magic;
	uint64_t n;                 // fragments of the part (in name order)
	uint64_t cigar_words, sequence_bytes, names_bytes;
	uint64_t windows, n_contigs; // coverage_t windows of the assembly, contigs (equal on all parts)
	uint64_t groups;            // read names of the part (group ids are 0 .. groups-1)
	uint64_t records, mapped_reads, malformed_count, missing_hi_tag, no_chimeric_reads, names_were_sorted, stream_bytes, max_read_length;
	uint64_t total_bytes;       // of the block, header included
	uint64_t qname_runs;        // runs of records with one read name in the stream of the part (SHARD_QNAME_KEYS: 16 bytes each)
	uint64_t reserved[6];
};

enum ShardSection {
	SHARD_N_ALN = 0, SHARD_FBITS, SHARD_GROUP,
	SHARD_SLOT0,                                   // per alignment slot: contig, start, end, abits, cigar_offset, cigar_count
	SHARD_SEQ_OFFSET0 = SHARD_SLOT0 + 18, SHARD_SEQ_LENGTH0, SHARD_SEQ_OFFSET1, SHARD_SEQ_LENGTH1,
	SHARD_CIGAR_POOL, SHARD_SEQ_POOL, SHARD_NAME_OFFSET, SHARD_NAMES,
	SHARD_WINDOWS32, SHARD_FRAGMENT_STARTS, SHARD_FRAGMENT_ENDS, SHARD_VIRAL_COUNTS, SHARD_QNAME_KEYS,
	SHARD_SECTIONS
};
enum { SHARD_SLOT_CONTIG = 0, SHARD_SLOT_START, SHARD_SLOT_END, SHARD_SLOT_ABITS, SHARD_SLOT_CIGAR_OFFSET, SHARD_SLOT_CIGAR_COUNT, SHARD_SLOT_FIELDS };

struct ShardLayout {
	uint64_t offset[SHARD_SECTIONS], bytes[SHARD_SECTIONS], total;
};

// where the sections of a part with these sizes lie inside its block
inline ShardLayout shard_layout(const ShardHeader& h) {
	ShardLayout layout;
	const uint64_t n = h.n;
	uint64_t* b = layout.bytes;
	b[SHARD_N_ALN] = n; b[SHARD_FBITS] = n; b[SHARD_GROUP] = n * 4;
	static const uint64_t field_bytes[SHARD_SLOT_FIELDS] = { 2, 4, 4, 1, 4, 2 };
	for (int slot = 0; slot < 3; ++slot)
		for (int field = 0; field < SHARD_SLOT_FIELDS; ++field) b[SHARD_SLOT0 + slot * SHARD_SLOT_FIELDS + field] = n * field_bytes[field];
	b[SHARD_SEQ_OFFSET0] = b[SHARD_SEQ_LENGTH0] = b[SHARD_SEQ_OFFSET1] = b[SHARD_SEQ_LENGTH1] = n * 4;
	b[SHARD_CIGAR_POOL] = h.cigar_words * 4; b[SHARD_SEQ_POOL] = h.sequence_bytes; b[SHARD_NAME_OFFSET] = (n + 1) * 8; b[SHARD_NAMES] = h.names_bytes;
	b[SHARD_WINDOWS32] = h.windows * 4; b[SHARD_FRAGMENT_STARTS] = h.windows; b[SHARD_FRAGMENT_ENDS] = h.windows; b[SHARD_VIRAL_COUNTS] = h.n_contigs * 8; b[SHARD_QNAME_KEYS] = h.qname_runs * 16;
	uint64_t at = sizeof(ShardHeader);
	for (int section = 0; section < SHARD_SECTIONS; ++section) { layout.offset[section] = at; at += (b[section] + 15) & ~(uint64_t) 15; }
	layout.total = at;
	return layout;
}

// what the parts add up to; returns an error text or null
inline const char* shard_totals(const ShardHeader* parts, uint32_t n_parts, ShardHeader& total) {
	memset(&total, 0, sizeof(total));
	total.magic = SHARD_MAGIC; total.names_were_sorted = 1; total.no_chimeric_reads = 1;
	for (uint32_t r = 0; r < n_parts; ++r) {
		const ShardHeader& h = parts[r];
		if (h.magic != SHARD_MAGIC) return "a block of the exchange does not start with a part of a sample (agpu_shard_export)";
		if (r > 0 && (h.windows != parts[0].windows || h.n_contigs != parts[0].n_contigs)) return "the parts of the sample were read against different assemblies";
		if (shard_layout(h).total != h.total_bytes) return "a part of the sample is damaged (its sizes do not add up)";
		total.n += h.n; total.cigar_words += h.cigar_words; total.sequence_bytes += h.sequence_bytes; total.names_bytes += h.names_bytes; total.groups += h.groups;
		total.qname_runs += h.qname_runs;
		total.records += h.records; total.mapped_reads += h.mapped_reads; total.malformed_count += h.malformed_count; total.missing_hi_tag += h.missing_hi_tag; total.stream_bytes += h.stream_bytes;
		if (!h.no_chimeric_reads) total.no_chimeric_reads = 0;
		if (!h.names_were_sorted) total.names_were_sorted = 0;
		if (h.max_read_length > total.max_read_length) total.max_read_length = h.max_read_length;
	}
	total.windows = n_parts ? parts[0].windows : 0; total.n_contigs = n_parts ? parts[0].n_contigs : 0;
	if (total.n >= 0xFFFFFFF0ull || total.qname_runs >= 0xFFFFFFF0ull) return "a batch holds at most 2^32-16 fragments";
	if (total.cigar_words >= 0xFFFFFFFFull || total.sequence_bytes / 4 >= 0xFFFFFFFFull) return "batch too large for 32-bit pool offsets"; // (the names have 64-bit offsets)
	return nullptr;
}

}

#endif
