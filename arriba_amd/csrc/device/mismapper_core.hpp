// arriba_amd/csrc/device/mismapper_core.hpp -- re-alignment of chimeric reads to the "other" gene
// (reference: make_kmer_index, align, align_both_strands, extend_split_read, filter_mismappers; source/filter_mismappers.cpp:16-359).
//
// The reference's k-mer index (unordered_map<8-mer, sorted vector<position>> per contig) is a CSR here: for every contig that
// holds an indexed gene a table of 4^8 + 1 offsets into one array of ascending positions.  The downstream splice sites of every
// gene (source/filter_mismappers.cpp:16-31) are a second CSR.  align() is the reference's greedy seed-and-extend search with its
// two recursive re-seeds; the recursion is an explicit stack of resumable frames, the order of evaluation is the reference's.
#ifndef AGPU_MISMAPPER_CORE_HPP
#define AGPU_MISMAPPER_CORE_HPP 1

#include <math.h>
#include "fusion_core.hpp"

namespace agpu {

const int KMER_LENGTH = 8;                 // source/arriba.cpp:548
const uint32_t KMER_COUNT = 1u << (2 * KMER_LENGTH);
const uint32_t NO_KMER_TABLE = 0xFFFFFFFFu;

struct KmerIndexView {
	const uint32_t* contig_table;   // [n_contigs] slot of the contig's offset table or NO_KMER_TABLE
	const uint32_t* offsets;        // [n_tables * KMER_COUNT + 1] into positions
	const int32_t* positions;       // ascending within a (contig, k-mer) bucket
	uint32_t n_contigs;
};
struct SpliceSiteView {
	const uint32_t* offset;         // [n_genes + n_dummy + 1]
	const int32_t* sites;           // downstream splice sites of the gene, ascending
	// a bit per position of the genome (bit genome.contig_offset[contig] + position): set where ANY gene has a downstream splice site, or null.  A walk that finds no bit
	// set over the positions it can reach has no splice site of its gene to look for and skips the binary search over the gene's sites (eight dependent look-ups per seed)
	const uint32_t* bits = nullptr;
};

// reference: kmer_to_int (source/filter_mismappers.cpp:33-45) on the genome's ASCII bases: T=0, G=1, C=2, everything else 3
AGPU_HD uint32_t kmer_digit_of_char(char base) { return base == 'T' ? 0u : base == 'G' ? 1u : base == 'C' ? 2u : 3u; }

// a substring of a read's sequence, optionally reverse-complemented (dna_to_reverse_complement, source/assembly.cpp)
struct Segment {
	SequenceRef sequence; uint32_t offset, length; bool reverse_complement;
	// optional copy of the segment's base characters as code(i) would give them (strand applied), element i at cache[i * cache_stride]: the search reads every base
	// of the segment hundreds of times; on the device the copy lives in LDS
	const uint8_t* cache = nullptr; uint32_t cache_stride = 1;
	// ... and as CHARACTERS, one byte per base, eight zero bytes behind the last (round 6): the walks compare eight bases of the read with eight of the gene in one XOR, and
	// code -> character cost more instructions than the comparison (chars8 was ~85 instructions, now one unaligned 8-byte read of LDS)
	const uint8_t* chars = nullptr;
	AGPU_HD uint32_t code(uint32_t i) const {
		if (cache != nullptr) return cache[(size_t) i * cache_stride];
		return reverse_complement ? complement_code(sequence.code(offset + length - 1 - i)) : sequence.code(offset + i);
	}
	AGPU_HD char at(uint32_t i) const { return chars != nullptr ? (char) chars[i] : base_char(code(i)); }
	// eight bases from position i on as characters, one per byte, the first in the low byte (0 behind the end of the segment): eight independent loads instead of
	// one per step of the extension
	AGPU_HD uint64_t chars8(uint32_t i) const {
		uint64_t value = 0;
		if (chars != nullptr) { __builtin_memcpy(&value, chars + i, 8); return value; }
		for (uint32_t k = 0; k < 8; ++k) if (i + k < length) value |= (uint64_t) (uint8_t) at(i + k) << (8 * k);
		return value;
	}
	AGPU_HD uint32_t kmer(uint32_t position) const {
		uint32_t result = 0;
		for (int k = 0; k < KMER_LENGTH; ++k) result = result << 2 | kmer_digit(code(position + k));
		return result;
	}
};

struct AlignTarget { // one gene window on one contig
	const char* contig_bases; // bases of the contig (genome.bases + contig offset)
	const uint32_t* kmer_offsets; // the contig's offset table (KMER_COUNT + 1 entries) or null
	const int32_t* positions;
	const int32_t* splice_sites; uint32_t n_splice_sites;
	int32_t gene_start, gene_end;
	const uint32_t* splice_bits = nullptr; uint64_t splice_bit_base = 0; // SpliceSiteView::bits and the bit of position 0 of the contig
};
// is a bit set among bits [first, last] of a bitmap?
AGPU_HD bool any_bit_in_range(const uint32_t* bits, uint64_t first, uint64_t last) {
	uint32_t found = 0;
	for (uint64_t word = first >> 5; word <= (last >> 5); ++word) {
		uint32_t mask = 0xFFFFFFFFu;
		if (word == (first >> 5)) mask &= 0xFFFFFFFFu << (first & 31);
		if (word == (last >> 5)) mask &= 0xFFFFFFFFu >> (31 - (last & 31));
		found |= bits[word] & mask;
	}
	return found != 0;
}

AGPU_HD uint32_t lower_bound_i32(const int32_t* values, uint32_t lo, uint32_t hi, int32_t value) {
	while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (values[mid] < value) lo = mid + 1; else hi = mid; }
	return lo;
}
AGPU_HD bool is_splice_site(const AlignTarget& target, int32_t position) {
	uint32_t at = lower_bound_i32(target.splice_sites, 0, target.n_splice_sites, position);
	return at < target.n_splice_sites && target.splice_sites[at] == position;
}
// the same question asked for ascending positions: `cursor` remembers where the last answer was found (one compare instead of a binary search per base)
AGPU_HD bool is_splice_site_from(const AlignTarget& target, int32_t position, uint32_t& cursor) {
	while (cursor < target.n_splice_sites && target.splice_sites[cursor] < position) ++cursor;
	return cursor < target.n_splice_sites && target.splice_sites[cursor] == position;
}
AGPU_HD uint64_t load_bases8(const char* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; } // eight genome bases at once (the genome buffer is padded behind its end)

// Memo of failed nested calls.  A nested align(score, read_pos, gene_pos, max_deletions) is monotone in `score` -- a higher score only loosens the bound of its
// read-position loop and raises every score it compares with min_score -- so a call that failed with score s fails with every score <= s.  The reference
// re-runs such calls from scratch: behind every splice site an extension crosses it starts a full search of the rest of the read against the rest of the gene,
// whose extensions cross splice sites again -- in a long gene with many exons the number of calls grows exponentially with the nesting (seconds to hours per
// read), although there are only (read positions x splice sites + seeds) distinct calls.  With the memo every distinct call is searched once per score level.
// The result is the reference's: only calls that are known to return false are skipped.
// A slot holds epoch (14 bits: one per align() invocation, so that the table is cleared once in 16 383 searches.  8 bits until round 5: a clear of the 8 MB table of a workgroup
// every 255 searches was 0.9 TB of writes per 10^8-fragment sample -- 29 M searches / 255 x 8 MB --, most of what the counters saw this kernel write) | gene_pos - gene_start (24) | read_pos (9) | max_deletions (1) |
// score + 32768 (16): for equal keys the larger word is the higher failed score.
const uint32_t ALIGN_MEMO_EPOCHS = 0x3FFFu; // the epoch field of a slot: bits 50-63
// The FRONT of the memo (round 6): a search lists five calls on average (1.46x10^8 calls of 2.9x10^7 searches at 10^8 fragments), and every walk that meets a mismatch asks the memo --
// two round trips to HBM in a chain of six (profiles/r06c_mismapper_times.txt; the kernel's time goes with the number of wavefronts in flight, profiles/r06e_heavy_ab.txt: it waits).
// The first keys of a search go into a small table in the memory the lanes share (LDS), the same slot format, the same epochs; a key that finds the ALIGN_MEMO_FRONT_PROBES slots
// of its probe sequence taken by other keys of the search goes to the table in HBM and says so in *spilled, and only then does a look-up that misses the front go on to HBM.  A slot
// of the front is never given back inside a search, so a key is in one of the two tables for good.
const int ALIGN_MEMO_FRONT_PROBES = 8;
struct AlignMemo {
	unsigned long long* slots; uint32_t mask; uint32_t epoch; // (no default initialisers: the device keeps one in LDS)
	unsigned long long* front; uint32_t front_mask; uint32_t* spilled; // null / 0 / null: no front table
	// (the key holds 24 bits of gene offset and 9 bits of read position: longer genes and reads are searched without the memo)
	AGPU_HD bool usable(int32_t gene_start, int32_t gene_end, int32_t read_length) const { return slots != nullptr && (int64_t) gene_end - gene_start < (1 << 24) && read_length < 512; }
	AGPU_HD unsigned long long key_of(int32_t read_pos, int32_t gene_offset, int32_t max_deletions, uint32_t strand = 0) const { // (the epochs of the searches are even: the odd one behind belongs to the reverse strand of a sweep over both)
		return ((unsigned long long) ((epoch + strand) & ALIGN_MEMO_EPOCHS) << 34 | (unsigned long long) (uint32_t) gene_offset << 10 | (unsigned long long) (uint32_t) read_pos << 1 | (unsigned long long) (max_deletions > 0)) << 16;
	}
	AGPU_HD uint32_t slot_of(unsigned long long key) const { unsigned long long h = key * 0x9E3779B97F4A7C15ull; return (uint32_t) (h >> 40) & mask; }
	// is a call with this key and a score <= the recorded one known to fail?
	AGPU_HD uint32_t front_slot_of(unsigned long long key) const { unsigned long long h = key * 0x9E3779B97F4A7C15ull; return (uint32_t) (h >> 28) & front_mask; }
	AGPU_HD bool known_to_fail(unsigned long long key, int32_t score) const {
		if (front != nullptr) {
			uint32_t at = front_slot_of(key);
			for (int probe = 0; probe < ALIGN_MEMO_FRONT_PROBES; ++probe, at = (at + 1) & front_mask) {
#if defined(__HIP_DEVICE_COMPILE__)
				const unsigned long long slot = __hip_atomic_load(&front[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
				const unsigned long long slot = front[at];
#endif
				if ((slot >> 16) == (key >> 16)) return score + 32768 <= (int32_t) (slot & 0xFFFF);
				if ((slot >> 50) != (key >> 50)) return false; // a free slot in its probe sequence: the key was never recorded (it would have taken this slot or one in front of it)
			}
#if defined(__HIP_DEVICE_COMPILE__)
			if (__hip_atomic_load(spilled, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) return false;
#else
			if (*spilled == 0) return false;
#endif
		}
		uint32_t at = slot_of(key);
		for (int probe = 0; probe < 16; ++probe, at = (at + 1) & mask) {
#if defined(__HIP_DEVICE_COMPILE__)
			const unsigned long long slot = __hip_atomic_load(&slots[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
			const unsigned long long slot = slots[at];
#endif
			if ((slot >> 16) == (key >> 16)) return score + 32768 <= (int32_t) (slot & 0xFFFF);
			if ((slot >> 50) != (key >> 50)) return false; // empty or from an earlier align(): the key is not in the table
		}
		return false;
	}
	AGPU_HD void record_failure(unsigned long long key, int32_t score) const {
		if (score < -32768 || score > 32767) return;
		const unsigned long long word = key | (unsigned long long) (uint32_t) (score + 32768);
		if (front != nullptr) {
			uint32_t at = front_slot_of(key);
			for (int probe = 0; probe < ALIGN_MEMO_FRONT_PROBES; ++probe, at = (at + 1) & front_mask) {
#if defined(__HIP_DEVICE_COMPILE__)
				const unsigned long long slot = __hip_atomic_load(&front[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				if ((slot >> 16) == (key >> 16)) { atomicMax(&front[at], word); return; }
				if ((slot >> 50) != (key >> 50)) {
					const unsigned long long seen = atomicCAS(&front[at], slot, word);
					if (seen == slot) return;
					if ((seen >> 16) == (key >> 16)) { atomicMax(&front[at], word); return; }
				}
#else
				const unsigned long long slot = front[at];
				if ((slot >> 16) == (key >> 16)) { if (word > slot) front[at] = word; return; }
				if ((slot >> 50) != (key >> 50)) { front[at] = word; return; }
#endif
			}
#if defined(__HIP_DEVICE_COMPILE__)
			__hip_atomic_store(spilled, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
			*spilled = 1;
#endif
		}
		uint32_t at = slot_of(key);
		for (int probe = 0; probe < 16; ++probe, at = (at + 1) & mask) {
#if defined(__HIP_DEVICE_COMPILE__)
			unsigned long long slot = __hip_atomic_load(&slots[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if ((slot >> 16) == (key >> 16)) { atomicMax(&slots[at], word); return; }
			if ((slot >> 50) != (key >> 50)) { // free for this epoch: take it (if another lane was faster with another key, go on probing)
				const unsigned long long seen = atomicCAS(&slots[at], slot, word);
				if (seen == slot) return;
				if ((seen >> 16) == (key >> 16)) { atomicMax(&slots[at], word); return; }
			}
#else
			const unsigned long long slot = slots[at];
			if ((slot >> 16) == (key >> 16)) { if (word > slot) slots[at] = word; return; }
			if ((slot >> 50) != (key >> 50)) { slots[at] = word; return; }
#endif
		}
	}
};

// One invocation of the reference's align() (source/filter_mismappers.cpp:86-199) as a resumable frame: the two recursive
// re-seeds of the reference push a child frame and continue behind the call when the child reports failure.
enum { ALIGN_NEXT_READ_POSITION = 0, ALIGN_NEXT_HIT = 1, ALIGN_RIGHT_LOOP = 2, ALIGN_COMPARE_BASE = 3, ALIGN_AFTER_MISMATCH = 4, ALIGN_ADVANCE = 5 };
struct AlignFrame {
	int32_t score, read_pos, skipped_bases, gene_pos, max_deletions;
	uint32_t hit, hits_end;               // current / end index into positions for the k-mer at read_pos
	int32_t extended_score, extended_read_pos, extended_gene_pos;
	uint32_t mismatch_count, consecutive_mismatches;
	uint32_t splice_cursor;               // index of the first splice site of the gene at or behind extended_gene_pos - 1 (the extension only moves forward)
	uint8_t leading;                      // read_pos == skipped_bases (only the outermost call starts at read position 0)
	uint8_t state;
	uint8_t started;                      // the read_pos loop has been entered (its increment runs before every later iteration)
};
const int ALIGN_MAX_DEPTH = 40;           // every nested call starts >= 8 bases further into a read of < 300 bases
const int ALIGN_SHALLOW_DEPTH = 16;       // ... of <= 128 bases: the stack of the first pass on the device

// The search as a list of independent pieces of work.  align() returns true as soon as ANY attempt anywhere in its recursion reaches min_score, and false when
// all of them are exhausted: its result is the OR over a tree of attempts, and the only thing the recursion adds is an order.  So a nested call need not be run
// where it is made: it is put on a list ("task": score, read position, gene position, deletions allowed) and the caller goes on at once as if the call had
// failed; whoever has nothing to do takes the next task.  Calls that the memo has seen with at least the same score are not listed again (whether the earlier
// one has finished or not: if it succeeds the answer is true anyway, if it fails so would this one).  One search of ~10^6 dependent steps in one lane -- the
// end of the second pass on the device waits for exactly that -- becomes rounds of up to 64 tasks, one per lane.  A list that overflows is abandoned and the
// search is done by the recursion.
// What the list holds are CALLS of align(): (score, read position, gene position) at the entry of the call, ALIGN_TASK_DELETIONS = max_deletions > 0, ALIGN_TASK_ROOT = the outermost
// call (its skipped bases are leading ones and cost nothing); from bit ALIGN_TASK_ITERATIONS_SHIFT on: the number of iterations of its read-position loop (align_by_sweep).
enum { ALIGN_TASK_DELETIONS = 1, ALIGN_TASK_ROOT = 2, ALIGN_TASK_STRAND = 4 /* a sweep over both strands of a segment: the call belongs to the search of the reverse complement */, ALIGN_TASK_ITERATIONS_SHIFT = 8 };
struct AlignTask { int32_t score, read_pos, gene_pos; uint32_t flags; };
// iterations of the read-position loop of a call: for (i = 0; read_pos + i + k < length && read_pos + i + min_score <= length + (score - i) + 2 k; ++i)
AGPU_HD uint32_t align_iterations(const AlignTask& call, int32_t length, int32_t min_score) {
	const int32_t by_length = length - KMER_LENGTH - call.read_pos;                                   // i < by_length
	const int32_t slack = length + call.score + 2 * KMER_LENGTH - call.read_pos - min_score;         // 2 i <= slack
	if (by_length <= 0 || slack < 0) return 0;
	const int32_t by_score = slack / 2 + 1;
	return (uint32_t) (by_length < by_score ? by_length : by_score);
}
// The search of one align() as ONE SWEEP over the read positions (round 3; AlignRunner::align_by_sweep).  A call listed by the extension of a seed at read position p starts at
// read position p + 8 or later, so when the sweep reaches a block of 8 read positions every call that reaches into the block is known.  For every seed of the block -- a hit of the
// 8-mer at its read position inside the gene -- the calls that reach it are compared: iteration i = read position - start of the call arrives with score - i + 8 + what the
// extension to the left over the i skipped bases adds; only the best arrival (per max_deletions) is walked to the right, because the walk itself does not depend on the score it
// starts with: a lower score finds no success a higher one does not find, and lists the same nested calls with lower scores.  Every seed of the gene is walked once (twice if
// the best arrival without deletions beats the best one with), where the recursion of the reference -- and the task list of round 2 -- walks it once per call that reaches it:
// 10^7 times for a read in a gene of some megabases (profiles/r03c_mismapper_second_pass.txt).
const uint32_t ALIGN_SWEEP_BLOCK = KMER_LENGTH;     // read positions per block
const uint32_t ALIGN_SWEEP_LOOKUPS = 64;           // read positions whose seeds are looked up at a time (a multiple of the block)
const uint32_t ALIGN_SWEEP_SEGMENT = 304;          // align_both_strands leaves segments of 300 bases and more alone
const uint32_t ALIGN_SWEEP_CALLS = 128;            // calls of a block kept in the memory the lanes share; what is beyond goes to AlignWorklist::relevant_words (256 until round 6: the 2 KB are
                                                   // the front of the memo and the head of the task list now)
const uint32_t ALIGN_MEMO_FRONT_SLOTS = 128;       // 1 KB: the front of the memo (AlignMemo::front)
const uint32_t ALIGN_LIST_HEAD_TASKS = 64;         // 1 KB: the head of the task list (AlignWorklist::head)
struct AlignSweep { // memory the lanes of a runner share (LDS on the device)
	uint32_t hit_first[2 * ALIGN_SWEEP_LOOKUPS], hit_count[2 * ALIGN_SWEEP_LOOKUPS]; // [strand][read position - first one of the look-ups]: the hits of its 8-mer inside the gene, as a range of the position list
	uint32_t seed_end[2 * ALIGN_SWEEP_BLOCK];                                 // running sums of the seeds of the read positions of the block, strand by strand
	int32_t call_score[ALIGN_SWEEP_CALLS], call_read_pos[ALIGN_SWEEP_CALLS], call_gene_pos[ALIGN_SWEEP_CALLS]; uint32_t call_flags[ALIGN_SWEEP_CALLS];
	uint32_t n_calls, reached;                                                // calls that reach into the block; bit 8 * strand + k: read position k of the block is reached by one of them
	uint32_t n_calls_forward;                                                 // the calls of the forward strand are the first of them (a seed looks at the calls of its strand only)
	uint32_t max_reach;                                                       // the first read position no listed call reaches (round 6: three blocks of five see no call at all -- the sweep ends there)
};
struct AlignWorklist {
	unsigned long long* words; uint32_t capacity; // two 64-bit words per task
	uint32_t* state;                              // [0] tasks listed, [1] overflow, [2] found (memory the lanes of the runner share: LDS on the device)
	AlignSweep* sweep;                            // null: the schedule of round 2 (the lanes take whole calls from the list in rounds)
	unsigned long long* relevant_words; uint32_t relevant_capacity; // the calls of a block beyond ALIGN_SWEEP_CALLS (two words per call; may be null: such a search is left to the recursion)
	unsigned long long* head; uint32_t head_capacity; // null / 0, or: the first tasks of the list ALSO in the memory the lanes share (round 6: the calls of every block are collected from the whole
	                                              // list -- one round trip to HBM per block of 8 read positions, 18 % of the wavefront time of the kernel -- and the list of most searches is short)
	uint32_t* stats;                              // null, or a study: [0] calls listed, [1] calls that reached into a block (summed over the blocks), [2] seeds, [3] walks, shared by the lanes;
	                                              // [4..6] ticks of the 100 MHz clock in the look-ups of the seeds / the collection of the calls of the blocks / the seeds (lane 0's view)
	AGPU_HD void push(const AlignTask& task) const {
#if defined(__HIP_DEVICE_COMPILE__)
		const uint32_t at = atomicAdd(&state[0], 1u);
#else
		const uint32_t at = state[0]++;
#endif
		if (at >= capacity) { state[1] = 1; return; }
		if (sweep != nullptr) {
			const uint32_t reach = (uint32_t) task.read_pos + (task.flags >> ALIGN_TASK_ITERATIONS_SHIFT);
#if defined(__HIP_DEVICE_COMPILE__)
			atomicMax(&sweep->max_reach, reach);
#else
			if (reach > sweep->max_reach) sweep->max_reach = reach;
#endif
		}
		const unsigned long long first = (unsigned long long) (uint32_t) task.score | (unsigned long long) (uint32_t) task.read_pos << 32, second = (unsigned long long) (uint32_t) task.gene_pos | (unsigned long long) task.flags << 32;
#if defined(__HIP_DEVICE_COMPILE__)
		__hip_atomic_store(&words[2 * (size_t) at], first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&words[2 * (size_t) at + 1], second, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (at < head_capacity) { __hip_atomic_store(&head[2 * (size_t) at], first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_store(&head[2 * (size_t) at + 1], second, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#else
		words[2 * (size_t) at] = first; words[2 * (size_t) at + 1] = second;
		if (at < head_capacity) { head[2 * (size_t) at] = first; head[2 * (size_t) at + 1] = second; }
#endif
	}
	AGPU_HD AlignTask task(uint32_t at) const {
#if defined(__HIP_DEVICE_COMPILE__)
		unsigned long long first, second;
		if (at < head_capacity) { first = __hip_atomic_load(&head[2 * (size_t) at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); second = __hip_atomic_load(&head[2 * (size_t) at + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
		else { first = __hip_atomic_load(&words[2 * (size_t) at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); second = __hip_atomic_load(&words[2 * (size_t) at + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
		const unsigned long long first = at < head_capacity ? head[2 * (size_t) at] : words[2 * (size_t) at], second = at < head_capacity ? head[2 * (size_t) at + 1] : words[2 * (size_t) at + 1];
#endif
		AlignTask result = { (int32_t) (uint32_t) first, (int32_t) (uint32_t) (first >> 32), (int32_t) (uint32_t) second, (uint32_t) (second >> 32) };
		return result;
	}
};

AGPU_HD void align_enter(AlignFrame& f, int32_t score, int32_t read_pos, int32_t gene_pos, int32_t max_deletions) {
	f.score = score; f.read_pos = read_pos; f.skipped_bases = 0; f.gene_pos = gene_pos; f.max_deletions = max_deletions;
	f.leading = read_pos == 0; f.state = ALIGN_NEXT_READ_POSITION; f.started = 0; f.hit = 0; f.hits_end = 0; f.splice_cursor = 0;
}

// One iteration of the outermost read_pos loop of the reference's align(): the seeds at read position `first_read_pos` with everything
// that follows from them (extensions, nested re-seeds).  The outermost loop only ever skips bases (score = -read_pos, all skipped bases
// leading), and its bound is monotone in read_pos, so its iterations are independent attempts: align() succeeds iff one of them does.
// That is what lets a wavefront try 64 read positions at once.
// `budget` (may be null): steps this search may still take; when it runs out -- or when the nesting exceeds `max_depth` frames -- the search is abandoned with
// *budget < 0 (the result is then meaningless and the caller must not use it: AlignRunner::exhausted).  A few reads of repetitive sequence need 10^2..10^4
// times the steps of an ordinary read, and a scheduler wants to know them.
// The frame that is being worked on lives in registers; `stack` only holds the frames of the callers (written when a nested call starts, read when it fails).
// `hit_offset`, `hit_stride`: of the seeds at the first read position this call only follows number hit_offset, hit_offset + hit_stride, ... (every seed is an
// independent attempt, too: the lanes of a wavefront share the seeds of one read position among them when a read has many).
#if !defined(__HIP_DEVICE_COMPILE__) && defined(AGPU_ALIGN_STATS)
struct AlignStats { unsigned long long read_positions, hits, bases, calls, pruned, failures, depth_sum; };
static AlignStats g_align_stats;
#define ALIGN_STAT(field, amount) (g_align_stats.field += (amount))
#else
#define ALIGN_STAT(field, amount) ((void) 0)
#endif
AGPU_HD bool align_search(const Segment& read, const AlignTarget& target, int32_t min_score, AlignFrame* stack, int max_depth, const AlignTask& task, int64_t* budget, uint32_t hit_offset = 0, uint32_t hit_stride = 1,
                          const AlignMemo* memo = nullptr, const AlignWorklist* worklist = nullptr) {
	const int32_t length = (int32_t) read.length;
	const bool use_memo = memo != nullptr && memo->usable(target.gene_start, target.gene_end, length);
	const bool root = (task.flags & ALIGN_TASK_ROOT) != 0;
	int depth = 0;
	AlignFrame f;
	align_enter(f, task.score, task.read_pos, task.gene_pos, (task.flags & ALIGN_TASK_DELETIONS) ? 1 : 0);
	if (root) { f.skipped_bases = task.read_pos; f.leading = 1; } // iteration read_pos of the outermost loop: score -read_pos, all skipped bases leading
	f.extended_score = 0; f.extended_read_pos = 0; f.extended_gene_pos = 0; f.mismatch_count = 0; f.consecutive_mismatches = 0;
	while (true) {
		if (budget != nullptr && --*budget < 0) return false;
		bool call = false, fail = false; // start a nested align() from (extended_score, extended_read_pos, extended_gene_pos) / this align() returns false
		int32_t call_max_deletions = 0;
		switch (f.state) {
			case ALIGN_NEXT_READ_POSITION: { // for (; read_pos + k < length && ...; read_pos++, score--, skipped_bases++)
				ALIGN_STAT(read_positions, 1);
				if (f.started && depth == 0 && root) return false; // the other read positions of the outermost loop are other attempts
				if (f.started) { f.read_pos++; f.score--; f.skipped_bases++; }
				f.started = 1;
				if (!(f.read_pos + KMER_LENGTH < length && f.read_pos + min_score <= length + f.score + 2 * KMER_LENGTH)) { fail = true; break; }
				if (target.kmer_offsets == 0) break; // no k-mer index on this contig: every lookup misses
				uint32_t kmer = read.kmer((uint32_t) f.read_pos);
				uint32_t begin = target.kmer_offsets[kmer], end = target.kmer_offsets[kmer + 1];
				f.hit = lower_bound_i32(target.positions, begin, end, f.gene_pos);
				f.hits_end = end;
				f.state = ALIGN_NEXT_HIT;
				if (depth == 0) { // this caller's share of the seeds
					if (end - f.hit > hit_offset) f.hit += hit_offset; else f.hit = end;
					if (f.hit < f.hits_end) f.hit -= hit_stride; else f.state = ALIGN_NEXT_READ_POSITION; // ALIGN_NEXT_HIT pre-increments (unsigned wrap-around on purpose)
				} else if (f.hit < f.hits_end) --f.hit; else f.state = ALIGN_NEXT_READ_POSITION;
				break;
			}
			case ALIGN_NEXT_HIT: { // for (hit = lower_bound(gene_pos); hit != end && *hit < gene_end; ++hit)
				ALIGN_STAT(hits, 1);
				f.hit += depth == 0 ? hit_stride : 1u;
				if (!(f.hit < f.hits_end && target.positions[f.hit] < target.gene_end)) { f.state = ALIGN_NEXT_READ_POSITION; break; }
				const int32_t kmer_hit = target.positions[f.hit];
				f.extended_score = f.score + KMER_LENGTH;
				if (f.leading) f.extended_score += f.skipped_bases; // no penalty for leading skipped bases (local alignment)
				if (f.extended_score >= min_score) return true;
				// extend to the left over the skipped bases, one mismatch allowed
				int32_t left_read_pos = f.read_pos - 1, left_gene_pos = kmer_hit - 1;
				uint32_t left_mismatches = 0;
				while (left_read_pos >= f.read_pos - f.skipped_bases && left_gene_pos >= target.gene_start) {
					if (read.at((uint32_t) left_read_pos) == target.contig_bases[left_gene_pos]) {
						f.extended_score += f.leading ? 1 : 2;
						if (f.extended_score >= min_score) return true;
					} else {
						if (++left_mismatches > 1) break;
					}
					left_read_pos--; left_gene_pos--;
				}
				f.extended_read_pos = f.read_pos + KMER_LENGTH;
				f.extended_gene_pos = kmer_hit + KMER_LENGTH;
				f.mismatch_count = 0; f.consecutive_mismatches = 0;
				f.splice_cursor = lower_bound_i32(target.splice_sites, 0, target.n_splice_sites, f.extended_gene_pos - 1);
				f.state = ALIGN_RIGHT_LOOP;
				break;
			}
			case ALIGN_RIGHT_LOOP: { // while (extended_read_pos < length && extended_gene_pos <= gene_end)
				// The extension to the right base by base, in one go until something happens that needs the stack (a nested call) or ends the hit: per base a
				// compare against a register window of eight genome bases and a compare against the next splice site.  What the states ALIGN_COMPARE_BASE /
				// ALIGN_AFTER_MISMATCH / ALIGN_ADVANCE do one at a time (they remain as the places where a nested call returns to).
				// What a step waits for is memory, so everything it reads comes in eights or is remembered: the genome bases (a register window of eight), the bases of
				// the read (another one), the position of the next splice site (looked up again only when the extension has passed it).
				uint64_t window = 0; int32_t window_at = 0, window_end = 0; // genome bases [window_at, window_end) of the contig
				uint64_t read_window = 0; int32_t read_window_at = 0, read_window_end = 0; // bases [read_window_at, read_window_end) of the read
				int32_t next_site = -1; // the first splice site at or behind extended_gene_pos - 1 (INT32_MAX: none); -1: not looked up yet
				while (true) {
					if (budget != nullptr && --*budget < 0) return false;
					ALIGN_STAT(bases, 1);
					if (!(f.extended_read_pos < length && f.extended_gene_pos <= target.gene_end)) { f.state = ALIGN_NEXT_HIT; break; }
					if (next_site < f.extended_gene_pos - 1) {
						while (f.splice_cursor < target.n_splice_sites && target.splice_sites[f.splice_cursor] < f.extended_gene_pos - 1) ++f.splice_cursor;
						next_site = f.splice_cursor < target.n_splice_sites ? target.splice_sites[f.splice_cursor] : 0x7FFFFFFF;
					}
					if (next_site == f.extended_gene_pos - 1) { f.state = ALIGN_COMPARE_BASE; call = true; call_max_deletions = f.max_deletions; break; } // re-seed behind a splice site (spliced alignment)
					if (f.extended_gene_pos >= window_end || f.extended_gene_pos < window_at) { window_at = f.extended_gene_pos; window_end = window_at + 8; window = load_bases8(target.contig_bases + window_at); }
					if (f.extended_read_pos >= read_window_end || f.extended_read_pos < read_window_at) { read_window_at = f.extended_read_pos; read_window_end = read_window_at + 8; read_window = read.chars8((uint32_t) read_window_at); }
					const char reference_base = (char) (window >> (8 * (f.extended_gene_pos - window_at)));
					if ((char) (read_window >> (8 * (f.extended_read_pos - read_window_at))) == reference_base) {
						f.extended_score++;
						if (f.extended_score >= min_score) return true;
						f.consecutive_mismatches = 0;
					} else {
						f.mismatch_count++;
						if (f.mismatch_count == 1 && f.max_deletions > 0 && length >= 30) { f.state = ALIGN_AFTER_MISMATCH; call = true; call_max_deletions = f.max_deletions - 1; break; } // re-seed once after the first mismatch (deletion / intron)
						f.extended_score--;
						f.consecutive_mismatches++;
						if (f.consecutive_mismatches >= 4) { f.state = ALIGN_NEXT_HIT; break; }
					}
					f.extended_read_pos++; f.extended_gene_pos++;
				}
				break;
			}
			case ALIGN_COMPARE_BASE: {
				if (read.at((uint32_t) f.extended_read_pos) == target.contig_bases[f.extended_gene_pos]) {
					f.extended_score++;
					if (f.extended_score >= min_score) return true;
					f.consecutive_mismatches = 0;
					f.state = ALIGN_ADVANCE;
				} else {
					f.mismatch_count++;
					f.state = ALIGN_AFTER_MISMATCH;
					if (f.mismatch_count == 1 && f.max_deletions > 0 && length >= 30) { call = true; call_max_deletions = f.max_deletions - 1; } // re-seed once after the first mismatch (deletion / intron)
				}
				break;
			}
			case ALIGN_AFTER_MISMATCH: {
				f.extended_score--;
				f.consecutive_mismatches++;
				f.state = (f.consecutive_mismatches >= 4) ? ALIGN_NEXT_HIT : ALIGN_ADVANCE;
				break;
			}
			default: { // ALIGN_ADVANCE
				f.extended_read_pos++; f.extended_gene_pos++;
				f.state = ALIGN_RIGHT_LOOP;
				break;
			}
		}
		const bool wanted_call = call;
		if (call && use_memo && memo->known_to_fail(memo->key_of(f.extended_read_pos, f.extended_gene_pos - target.gene_start, call_max_deletions), f.extended_score)) call = false; // searched before, with at least this score
		if (wanted_call && !call) ALIGN_STAT(pruned, 1);
		if (call && worklist != nullptr) { // listed for whoever is free; this search goes on as if the call had failed (the result is an OR over everything that gets searched)
			if (use_memo) memo->record_failure(memo->key_of(f.extended_read_pos, f.extended_gene_pos - target.gene_start, call_max_deletions), f.extended_score); // (seen: not listed again with a score <= this one)
			const AlignTask nested = { f.extended_score, f.extended_read_pos, f.extended_gene_pos, call_max_deletions > 0 ? (uint32_t) ALIGN_TASK_DELETIONS : 0u };
			worklist->push(nested);
			ALIGN_STAT(calls, 1);
			call = false;
		}
		if (call) ALIGN_STAT(calls, 1); else if (fail) ALIGN_STAT(failures, 1);
		if (call) ALIGN_STAT(depth_sum, depth + 1);
		if (call) { // the caller's frame goes to the stack, the nested call takes the registers
			if (depth >= max_depth) { if (budget != nullptr) *budget = -1; return false; } // deeper than this stack: left to a caller with a deeper one
			stack[depth] = f;
			++depth;
			const int32_t score = f.extended_score, read_pos = f.extended_read_pos, gene_pos = f.extended_gene_pos;
			align_enter(f, score, read_pos, gene_pos, call_max_deletions);
		} else if (fail) { // this call returns false: the caller goes on behind the call
			if (depth == 0) return false;
			// (a nested call entered with read position read_pos - skipped_bases and score score + skipped_bases: both move together in its loop)
			if (use_memo) memo->record_failure(memo->key_of(f.read_pos - f.skipped_bases, f.gene_pos - target.gene_start, f.max_deletions), f.score + f.skipped_bases);
			--depth;
			f = stack[depth];
		}
	}
}

#if !defined(__HIP_DEVICE_COMPILE__)
static unsigned long long g_align_seed_steps = 0; // host stepping: bases compared by the walks of the sweep (the harness reports what a wavefront would wait for)
#define ALIGN_SEED_STEP() (++g_align_seed_steps)
#else
#define ALIGN_SEED_STEP() ((void) 0)
#endif
// The bases to the left of a seed at (read_pos, kmer_hit) as the extension to the left of the reference's align() sees them (source/filter_mismappers.cpp:109-137): the read is
// compared backwards with the gene, one mismatch is allowed, the second one ends the extension, and so do the start of the read and the start of the gene.  first / second =
// number of bases compared before the first / second mismatch (or `limit`, the number of bases there are, if it does not come).
struct LeftProfile { int32_t first, second, limit; };
AGPU_HD LeftProfile align_left_profile(const Segment& read, const AlignTarget& target, int32_t read_pos, int32_t kmer_hit) {
	LeftProfile profile;
	const int32_t by_gene = kmer_hit - target.gene_start; // left_gene_pos = kmer_hit - 1 - j >= gene_start
	profile.limit = read_pos < by_gene ? read_pos : by_gene;
	if (profile.limit < 0) profile.limit = 0;
	profile.first = profile.second = profile.limit;
	int32_t mismatches = 0;
	for (int32_t done = 0; done < profile.limit && mismatches < 2; done += 8) {
		const int32_t gene_from = kmer_hit - done; // the eight gene bases in front of it, the nearest in the high byte
		uint64_t window = 0;
		if (gene_from >= 8) window = load_bases8(target.contig_bases + gene_from - 8);
		else for (int32_t k = 0; k < gene_from; ++k) window |= (uint64_t) (uint8_t) target.contig_bases[gene_from - 1 - k] << (8 * (7 - k));
		if (read.chars != nullptr) { // the eight bases of the read in front of read_pos - done the same way, and the mismatches from the bytes that differ
			const int32_t read_from = read_pos - done; // (>= 1: done < limit <= read_pos)
			uint64_t mine;
			if (read_from >= 8) __builtin_memcpy(&mine, read.chars + read_from - 8, 8);
			else { __builtin_memcpy(&mine, read.chars, 8); mine <<= 8 * (8 - read_from); }
			uint64_t differ = window ^ mine;
			const int32_t valid = profile.limit - done < 8 ? profile.limit - done : 8;
			if (valid < 8) differ &= ~0ull << (8 * (8 - valid));
			ALIGN_SEED_STEP();
			while (differ != 0 && mismatches < 2) {
				const int32_t k = __builtin_clzll(differ) >> 3;
				if (++mismatches == 1) profile.first = done + k; else profile.second = done + k;
				differ &= ~(0xFFull << (8 * (7 - k)));
			}
			continue;
		}
		for (int32_t k = 0; k < 8 && done + k < profile.limit; ++k) {
			ALIGN_SEED_STEP();
			if (read.at((uint32_t) (read_pos - 1 - done - k)) == (char) (window >> (8 * (7 - k)))) continue;
			if (++mismatches == 1) profile.first = done + k; else { profile.second = done + k; break; }
		}
	}
	return profile;
}
// bases that match among the first i to the left (i skipped bases): those before the second mismatch, less the first mismatch
AGPU_HD int32_t align_left_matches(const LeftProfile& profile, int32_t i) {
	const int32_t compared = i < profile.second ? i : profile.second;
	return compared - (profile.first < compared ? 1 : 0);
}

// The extension to the right of a seed, to its end: the rest of the body of the hit loop of the reference's align() (source/filter_mismappers.cpp:139-183) with the nested calls LISTED
// instead of made (the search goes on as if they had failed: the result of align() is an OR over everything that gets searched, see AlignWorklist).  No stack, no state machine:
// what ALIGN_RIGHT_LOOP .. ALIGN_ADVANCE of align_search do for one hit.  extended_score = the score behind the extension to the left.
// What a walk needs first, asked for BEFORE the extension to the left is looked at (round 6): the eight gene bases behind the seed and whether a splice site is in reach of the walk.
// The walk of a seed waits for a chain of loads -- position list, gene bases to the left, splice bits, gene bases to the right, memo --; these two do not depend on the ones in
// front of them, and asked for together with the bases to the left they cost one round trip instead of three.  (The genome is padded by 16 bytes behind its last base.)
struct SeedAhead { uint64_t right_window; bool splice_site_in_reach; };
AGPU_HD SeedAhead align_seed_ahead(const Segment& read, const AlignTarget& target, int32_t read_pos, int32_t kmer_hit) {
	SeedAhead ahead;
	const int32_t extended_read_pos = read_pos + KMER_LENGTH, extended_gene_pos = kmer_hit + KMER_LENGTH;
	ahead.right_window = load_bases8(target.contig_bases + extended_gene_pos);
	ahead.splice_site_in_reach = target.splice_bits == nullptr || any_bit_in_range(target.splice_bits, target.splice_bit_base + (uint64_t) (extended_gene_pos - 1), target.splice_bit_base + (uint64_t) (extended_gene_pos - 1) + (uint64_t) ((int32_t) read.length - extended_read_pos));
	return ahead;
}
// A nested call of a walk, listed unless the memo knows one with at least its score (see AlignWorklist).
AGPU_HD void align_list_call(const AlignTarget& target, int32_t length, int32_t min_score, int32_t score, int32_t read_pos, int32_t gene_pos, int32_t call_max_deletions, const AlignMemo& memo, const AlignWorklist& worklist, uint32_t strand) {
	AlignTask nested = { score, read_pos, gene_pos, (call_max_deletions > 0 ? (uint32_t) ALIGN_TASK_DELETIONS : 0u) | (strand != 0 ? (uint32_t) ALIGN_TASK_STRAND : 0u) };
	const uint32_t iterations = align_iterations(nested, length, min_score);
	if (iterations == 0) return; // (a call whose loop does not run returns false at once)
	const unsigned long long key = memo.key_of(read_pos, gene_pos - target.gene_start, call_max_deletions, strand);
	if (memo.known_to_fail(key, score)) { ALIGN_STAT(pruned, 1); return; } // listed before with at least this score
	memo.record_failure(key, score);
	nested.flags |= iterations << ALIGN_TASK_ITERATIONS_SHIFT;
	worklist.push(nested);
	ALIGN_STAT(calls, 1);
}
// The walk(s) of a seed to the right in ONE pass (round 6).  A seed is walked with the best arrival that may still delete (max_deletions 1: `score_with`) and, if an arrival that
// may not is better, with that one as well (`score_without`); ALIGN_NO_ARRIVAL = no such walk.  The two walks compare the same bases, so their scores differ by a constant all the
// way and they end at the same base: one pass with the higher score in the lead lists what both would list -- behind a splice site a call of each, after the first mismatch the
// re-seed of the walk that may still delete (source/filter_mismappers.cpp:147-171) -- and succeeds when the lead reaches min_score (the other one cannot be first).
// THE BOUND: a base of the read adds at most 1 to a score -- a match of a walk; a base a call skips and its seed then finds matching to the left costs 1 and gives 2 back; the eight of
// a seed give 8 -- so from (score, read position) nothing above score + (length - read position) is ever reached, by the walk or by anything it lists.  A walk whose lead has fallen
// below min_score - (bases left) is over, and a walker in that state lists nothing.  The reference walks on and lists calls that its own loop bound (:92, looser by 2 k) lets run for
// up to eight iterations of seeds without a chance: 43 % of the steps of the walks of tests/golden's stress sample.  Only failures are left out: the verdict is an OR over successes.
const int32_t ALIGN_NO_ARRIVAL = -0x40000000;
AGPU_HD bool align_walk_seed(const Segment& read, const AlignTarget& target, int32_t min_score, int32_t score_with, int32_t score_without, int32_t read_pos, int32_t kmer_hit, const AlignMemo& memo, const AlignWorklist& worklist, const SeedAhead* ahead = nullptr, uint32_t strand = 0) {
	const int32_t length = (int32_t) read.length;
	const bool with = score_with != ALIGN_NO_ARRIVAL, without = score_without != ALIGN_NO_ARRIVAL;
	ALIGN_STAT(hits, 1);
#if defined(__HIP_DEVICE_COMPILE__)
	if (worklist.stats != nullptr) atomicAdd(&worklist.stats[3], (with ? 1u : 0u) + (without ? 1u : 0u));
#endif
	int32_t lead = without ? score_without : score_with;     // (the caller passes score_without only if it is the higher one)
	const int32_t behind = with && without ? score_without - score_with : 0; // the walk that may delete: lead - behind
	if (lead >= min_score) return true;
	int32_t extended_read_pos = read_pos + KMER_LENGTH, extended_gene_pos = kmer_hit + KMER_LENGTH;
	uint32_t mismatch_count = 0, consecutive_mismatches = 0;
	// the first splice site at or behind extended_gene_pos - 1 -- if the walk can reach one at all: it compares positions extended_gene_pos - 1 ... + (length - extended_read_pos)
	uint32_t splice_cursor = target.n_splice_sites; int32_t next_site = 0x7FFFFFFF;
	if (ahead != nullptr ? ahead->splice_site_in_reach : (target.splice_bits == nullptr || any_bit_in_range(target.splice_bits, target.splice_bit_base + (uint64_t) (extended_gene_pos - 1), target.splice_bit_base + (uint64_t) (extended_gene_pos - 1) + (uint64_t) (length - extended_read_pos)))) {
		splice_cursor = lower_bound_i32(target.splice_sites, 0, target.n_splice_sites, extended_gene_pos - 1);
		next_site = splice_cursor < target.n_splice_sites ? target.splice_sites[splice_cursor] : 0x7FFFFFFF;
	}
	// bytes that differ between the genome bases [window_at, window_at + 8) of the contig and the bases of the read that stand against them (gene and read advance together)
	uint64_t differ = 0; int32_t window_at = extended_gene_pos - 8;
	bool fetched = ahead != nullptr;
	ALIGN_SEED_STEP();
	// The re-seed after the first mismatch is LISTED BEHIND THE LOOP: nearly every walk has one, each lane of a wavefront at a base of its own, and listing it where it arises made
	// the wavefront run through the listing (memo, list: ~150 instructions) once per distinct base instead of once.  (A walk that succeeds lists nothing: the search is over.)
	bool reseed = false; int32_t reseed_score = 0, reseed_read_pos = 0, reseed_gene_pos = 0;
	while (extended_read_pos < length && extended_gene_pos <= target.gene_end) {
		const int32_t short_of = min_score - (length - extended_read_pos); // a score below this one cannot reach min_score any more
		if (lead < short_of) break;
		ALIGN_STAT(bases, 1); ALIGN_SEED_STEP();
		if (next_site < extended_gene_pos - 1) {
			while (splice_cursor < target.n_splice_sites && target.splice_sites[splice_cursor] < extended_gene_pos - 1) ++splice_cursor;
			next_site = splice_cursor < target.n_splice_sites ? target.splice_sites[splice_cursor] : 0x7FFFFFFF;
		}
		if (next_site == extended_gene_pos - 1) { // behind a splice site, before the base is compared: a call of every walk, with its max_deletions
			if (with && lead - behind >= short_of) align_list_call(target, length, min_score, lead - behind, extended_read_pos, extended_gene_pos, 1, memo, worklist, strand);
			if (without) align_list_call(target, length, min_score, lead, extended_read_pos, extended_gene_pos, 0, memo, worklist, strand);
		}
		if (extended_gene_pos >= window_at + 8) {
			window_at = extended_gene_pos;
			differ = (fetched ? ahead->right_window : load_bases8(target.contig_bases + window_at)) ^ read.chars8((uint32_t) extended_read_pos);
			fetched = false;
		}
		if (((differ >> (8 * (extended_gene_pos - window_at))) & 0xFF) == 0) {
			lead++;
			if (lead >= min_score) return true;
			consecutive_mismatches = 0;
		} else {
			mismatch_count++;
			// the walk that may delete re-seeds once, after its first mismatch and before it is counted against the score (deletion / intron)
			if (mismatch_count == 1 && with && length >= 30 && lead - behind >= short_of) { reseed = true; reseed_score = lead - behind; reseed_read_pos = extended_read_pos; reseed_gene_pos = extended_gene_pos; }
			lead--;
			consecutive_mismatches++;
			if (consecutive_mismatches >= 4) break; // on to the next seed
		}
		extended_read_pos++; extended_gene_pos++;
	}
	if (reseed) align_list_call(target, length, min_score, reseed_score, reseed_read_pos, reseed_gene_pos, 0, memo, worklist, strand);
	return false;
}

AGPU_HD bool align_from_read_position(const Segment& read, const AlignTarget& target, int32_t min_score, AlignFrame* stack, int max_depth, int32_t first_read_pos, int64_t* budget, uint32_t hit_offset = 0, uint32_t hit_stride = 1, const AlignMemo* memo = nullptr) {
	const AlignTask task = { -first_read_pos, first_read_pos, target.gene_start, ALIGN_TASK_ROOT | ALIGN_TASK_DELETIONS };
	return align_search(read, target, min_score, stack, max_depth, task, budget, hit_offset, hit_stride, memo, nullptr);
}

// Who tries the read positions: one thread after the other on the host (lanes = 1), the 64 lanes of a wavefront on the device.
// SWEEP_ONLY: the wavefront-per-read kernel of the device is compiled twice -- once with nothing but the sweep (a search the sweep cannot hold -- a gene of 2^24 bases and more,
// lists that run over -- is given up with *budget = -1 and the read left to the second instantiation), once with everything: the recursion and its stack of frames need
// registers and scratch memory that the sweep does not, and the kernel that does almost all of the work should not carry them (agpu_mismappers.hip: mismapper_heavy_kernel).
template <bool SWEEP_ONLY> struct AlignRunnerT {
	AlignFrame* stack; uint32_t lane, lanes;
	int64_t* budget = nullptr; // steps left for the whole verdict of one read (null: unlimited)
	int max_depth = ALIGN_MAX_DEPTH; // frames `stack` holds
	uint8_t* cache = nullptr; uint32_t cache_stride = 1, cache_capacity = 0; // room for a copy of the segment being searched (LDS on the device), shared by the lanes of the runner
	uint8_t* chars = nullptr, * chars2 = nullptr;                            // ... and for the characters of both (cache_capacity + 8 bytes each; null: the searches translate the codes)
	uint8_t* cache2 = nullptr;                                               // ... and for its reverse complement (same stride and capacity): a sweep over both strands (align_strands)
	bool strands_together = true;                                            // (false: strand by strand as until round 6, for A/B measurements)
	AGPU_HD bool exhausted() const { return budget != nullptr && *budget < 0; }
	// the base characters of the segment, strand applied, where the search finds them fast
	AGPU_HD Segment prepared(const Segment& segment) const {
		Segment result = segment;
		if (cache == nullptr || segment.length > cache_capacity) return result;
#if defined(__HIP_DEVICE_COMPILE__)
		if (lanes > 1) __syncthreads(); // (lanes > 1: the lanes are one workgroup and run this code together; nobody still reads the previous copy)
#endif
		for (uint32_t i = lane; i < segment.length; i += lanes) cache[(size_t) i * cache_stride] = (uint8_t) segment.code(i);
		if (chars != nullptr) for (uint32_t i = lane; i < segment.length + 8; i += lanes) chars[i] = i < segment.length ? (uint8_t) base_char(segment.code(i)) : (uint8_t) 0;
#if defined(__HIP_DEVICE_COMPILE__)
		if (lanes > 1) __syncthreads();
#endif
		result.cache = cache; result.cache_stride = cache_stride; result.chars = chars;
		return result;
	}
	AGPU_HD bool any(bool mine) const {
#if defined(__HIP_DEVICE_COMPILE__)
		return lanes > 1 ? __any(mine) != 0 : mine;
#else
		return mine;
#endif
	}
	AGPU_HD void sync_lanes() const {
#if defined(__HIP_DEVICE_COMPILE__)
		if (lanes > 1) __syncthreads();
#endif
	}
	AGPU_HD void new_memo_epoch() const {
		sync_lanes();
		const uint32_t next = (memo->epoch + 2) & ALIGN_MEMO_EPOCHS; // (every lane reads the same value)
		if (next == 0) for (uint32_t k = lane; k <= memo->mask; k += lanes) memo->slots[k] = 0; // the epoch numbers wrap around: start clean (epoch 0 = empty slots)
		if (next == 0 && memo->front != nullptr) for (uint32_t k = lane; k <= memo->front_mask; k += lanes) memo->front[k] = 0;
		sync_lanes();
		if (lane == 0 && memo->front != nullptr) *memo->spilled = 0;
		if (lane == 0) memo->epoch = next == 0 ? 2 : next;
		sync_lanes();
	}
	// reference: align(0, read, 0, contig, gene_start, gene_start, gene_end, ...) (source/filter_mismappers.cpp:86-199)
	AlignMemo* memo = nullptr; // table of failed nested calls, shared by the lanes of the runner (second pass on the device); every align() takes a new epoch
	AlignWorklist* worklist = nullptr; // the search as a list of tasks the lanes take in rounds (see AlignWorklist); needs the memo
#if !defined(__HIP_DEVICE_COMPILE__)
	uint32_t virtual_lanes = 1;        // host stepping only: tasks per round, taken one after the other (what 64 lanes take at once on the device)
	unsigned long long* round_steps = nullptr; // host stepping only (with a budget): [0] += the steps of the longest task of every round, [1] += rounds
#endif
	bool lanes_share_seeds = false; // the lanes work on the same read position and split its seeds (reads with hundreds of seeds per position); default: one read position per lane
	// the lanes of a round: on the device every lane does its own share (lane, lane + 64, ...), the host steps through all of them
	AGPU_HD uint32_t first_of_mine(uint32_t width) const {
#if defined(__HIP_DEVICE_COMPILE__)
		return lane;
#else
		return 0;
#endif
	}
	AGPU_HD uint32_t stride_of_mine(uint32_t width) const {
#if defined(__HIP_DEVICE_COMPILE__)
		return lanes;
#else
		return 1;
#endif
	}
	// The calls of the list that reach into the block of read positions [block, block + 8): into the memory of the sweep (the first ALIGN_SWEEP_CALLS) and the overflow list.
	// Returns false if there are more of them than both hold.
	AGPU_HD bool sweep_collect_calls(AlignSweep& sweep, int32_t block, uint32_t listed, uint32_t width, uint32_t strands) const {
		uint32_t n = 0, reached = 0, n_forward = 0;
#if defined(__HIP_DEVICE_COMPILE__)
		if (lane == 0) sweep.reached = 0;
		sync_lanes();
		for (uint32_t strand = 0; strand < strands; ++strand) { // (strand by strand, so that the calls of a strand stand together)
		for (uint32_t base = 0; base < listed; base += lanes) {
			const uint32_t j = base + lane;
			AlignTask call = { 0, 0, 0, 0 };
			bool relevant = false;
			if (j < listed) {
				call = worklist->task(j);
				const int32_t iterations = (int32_t) (call.flags >> ALIGN_TASK_ITERATIONS_SHIFT);
				relevant = call.read_pos < block + (int32_t) ALIGN_SWEEP_BLOCK && call.read_pos + iterations > block && ((call.flags & ALIGN_TASK_STRAND) != 0) == (strand != 0);
			}
			const unsigned long long mask = __ballot(relevant);
			if (relevant) {
				const uint32_t at = n + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
				sweep_store_call(sweep, at, call);
			}
			if (relevant) atomicOr(&sweep.reached, reached_bits(call, block)); // which read positions of the block does one of these calls reach?
			n += (uint32_t) __popcll(mask);
		}
		if (strand == 0) n_forward = n;
		}
		sync_lanes();
		if (lane == 0) { sweep.n_calls = n; sweep.n_calls_forward = n_forward; }
		sync_lanes();
		(void) reached;
#else
		for (uint32_t strand = 0; strand < strands; ++strand) {
			for (uint32_t j = 0; j < listed; ++j) {
				const AlignTask call = worklist->task(j);
				const int32_t iterations = (int32_t) (call.flags >> ALIGN_TASK_ITERATIONS_SHIFT);
				if (!(call.read_pos < block + (int32_t) ALIGN_SWEEP_BLOCK && call.read_pos + iterations > block) || ((call.flags & ALIGN_TASK_STRAND) != 0) != (strand != 0)) continue;
				sweep_store_call(sweep, n++, call);
				reached |= reached_bits(call, block);
			}
			if (strand == 0) n_forward = n;
		}
		sweep.n_calls = n; sweep.n_calls_forward = n_forward; sweep.reached = reached;
#endif
		return n <= ALIGN_SWEEP_CALLS + (worklist->relevant_words != nullptr ? worklist->relevant_capacity : 0u);
	}
	// the read positions [block, block + 8) that the loop of a call stands at, as bits 0-7 (forward strand) or 8-15 (a call of the search of the reverse complement); the call reaches into the block
	AGPU_HD static uint32_t reached_bits(const AlignTask& call, int32_t block) {
		const int32_t iterations = (int32_t) (call.flags >> ALIGN_TASK_ITERATIONS_SHIFT);
		const int32_t from = call.read_pos > block ? call.read_pos - block : 0, to = call.read_pos + iterations < block + (int32_t) ALIGN_SWEEP_BLOCK ? call.read_pos + iterations - block : (int32_t) ALIGN_SWEEP_BLOCK;
		return (((1u << to) - 1u) & ~((1u << from) - 1u)) << ((call.flags & ALIGN_TASK_STRAND) ? ALIGN_SWEEP_BLOCK : 0u);
	}
	AGPU_HD void sweep_store_call(AlignSweep& sweep, uint32_t at, const AlignTask& call) const {
		if (at < ALIGN_SWEEP_CALLS) { sweep.call_score[at] = call.score; sweep.call_read_pos[at] = call.read_pos; sweep.call_gene_pos[at] = call.gene_pos; sweep.call_flags[at] = call.flags; return; }
		at -= ALIGN_SWEEP_CALLS;
		if (worklist->relevant_words == nullptr || at >= worklist->relevant_capacity) return; // (the caller sees the count)
		const unsigned long long first = (unsigned long long) (uint32_t) call.score | (unsigned long long) (uint32_t) call.read_pos << 32, second = (unsigned long long) (uint32_t) call.gene_pos | (unsigned long long) call.flags << 32;
#if defined(__HIP_DEVICE_COMPILE__)
		__hip_atomic_store(&worklist->relevant_words[2 * (size_t) at], first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&worklist->relevant_words[2 * (size_t) at + 1], second, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
		worklist->relevant_words[2 * (size_t) at] = first; worklist->relevant_words[2 * (size_t) at + 1] = second;
#endif
	}
	AGPU_HD AlignTask sweep_call(const AlignSweep& sweep, uint32_t at) const {
		if (at < ALIGN_SWEEP_CALLS) { const AlignTask call = { sweep.call_score[at], sweep.call_read_pos[at], sweep.call_gene_pos[at], sweep.call_flags[at] }; return call; }
		at -= ALIGN_SWEEP_CALLS;
#if defined(__HIP_DEVICE_COMPILE__)
		const unsigned long long first = __hip_atomic_load(&worklist->relevant_words[2 * (size_t) at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), second = __hip_atomic_load(&worklist->relevant_words[2 * (size_t) at + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
		const unsigned long long first = worklist->relevant_words[2 * (size_t) at], second = worklist->relevant_words[2 * (size_t) at + 1];
#endif
		const AlignTask call = { (int32_t) (uint32_t) first, (int32_t) (uint32_t) (first >> 32), (int32_t) (uint32_t) second, (uint32_t) (second >> 32) };
		return call;
	}
	// one seed of the block: the best arrival of the calls that reach it, walked to the right
	AGPU_HD bool sweep_seed(const AlignSweep& sweep, const Segment& read, const AlignTarget& target, int32_t min_score, int32_t read_pos, int32_t kmer_hit, uint32_t strand) const {
#if defined(__HIP_DEVICE_COMPILE__)
		const bool timed = worklist->stats != nullptr && lane == 0; // (the study of ARRIBA_MISMAPPER_TIMES: lane 0 has a seed in every round, and the lanes of a wavefront meet again behind every part)
		unsigned long long tick = timed ? wall_clock64() : 0ull;
#define SEED_LAP(slot) do { if (timed) { const unsigned long long now_ = wall_clock64(); worklist->stats[slot] += (uint32_t) (now_ - tick); tick = now_; } } while (0)
#else
#define SEED_LAP(slot) ((void) 0)
#endif
		const SeedAhead ahead = align_seed_ahead(read, target, read_pos, kmer_hit);
		const LeftProfile profile = align_left_profile(read, target, read_pos, kmer_hit);
		SEED_LAP(8);
		const int32_t NONE = -0x40000000;
		int32_t best[2] = { NONE, NONE }; // by max_deletions
		const uint32_t n_calls = strand != 0 ? sweep.n_calls : sweep.n_calls_forward;
		for (uint32_t c = strand != 0 ? sweep.n_calls_forward : 0u; c < n_calls; ++c) { // the calls of the seed's strand
			const AlignTask call = sweep_call(sweep, c);
			const int32_t i = read_pos - call.read_pos; // the iteration of the call's loop that stands at this read position: score - i, i skipped bases
			if (i < 0 || i >= (int32_t) (call.flags >> ALIGN_TASK_ITERATIONS_SHIFT) || call.gene_pos > kmer_hit) continue;
			const int32_t matches = align_left_matches(profile, i);
			// source/filter_mismappers.cpp:111-135: + k for the seed; leading skipped bases cost nothing and a match among them counts 1, otherwise it takes back the penalty and counts 1: 2
			const int32_t extended_score = call.score - i + KMER_LENGTH + ((call.flags & ALIGN_TASK_ROOT) ? i + matches : 2 * matches);
			const int deletions = (call.flags & ALIGN_TASK_DELETIONS) ? 1 : 0;
			if (extended_score > best[deletions]) best[deletions] = extended_score;
		}
		SEED_LAP(9);
		// (an arrival that may not delete with at most the score of one that may lists nothing new; one that cannot reach min_score over the bases behind the seed walks nowhere: align_walk_seed)
		const int32_t short_of = min_score - ((int32_t) read.length - read_pos - (int32_t) KMER_LENGTH);
		const int32_t score_with = best[1] != NONE && best[1] >= short_of ? best[1] : ALIGN_NO_ARRIVAL, score_without = best[0] != NONE && best[0] > best[1] && best[0] >= short_of ? best[0] : ALIGN_NO_ARRIVAL;
		const bool found = (score_with != ALIGN_NO_ARRIVAL || score_without != ALIGN_NO_ARRIVAL) && align_walk_seed(read, target, min_score, score_with, score_without, read_pos, kmer_hit, *memo, *worklist, &ahead, strand);
		SEED_LAP(10);
		return found;
	}
	// returns whether the segment aligns; worklist->state[1] != 0 afterwards: a list was too short, the answer is not known (left to the recursion)
	// strands = 2 (round 6): `forward` and its reverse complement (whose base codes the caller has put into cache2) in ONE sweep.  The two searches of align_both_strands
	// (source/filter_mismappers.cpp:218-224) run over the same read positions against the same gene; their calls share the list (ALIGN_TASK_STRAND), their seeds the rounds of a block:
	// a block of one search has ~22 seeds for 64 lanes, and the kernel is bound by the instructions it issues (profiles/r06f_sq_summary.txt) -- with two searches a round is twice
	// as full and there are half as many.  The reference does not search the reverse strand when the forward one aligns; searching it anyway changes nothing: the answer is an OR.
	AGPU_HD bool align_by_sweep(const Segment& forward, uint32_t strands, const AlignTarget& target, int32_t min_score) const {
		AlignSweep& sweep = *worklist->sweep;
		const int32_t length = (int32_t) forward.length;
#if !defined(__HIP_DEVICE_COMPILE__)
		const uint32_t width = virtual_lanes > 1 ? virtual_lanes : 1; // host stepping: the lanes of the device one after the other
#else
		const uint32_t width = lanes;
#endif
		sync_lanes();
		if (lane == 0) {
			worklist->state[0] = 0; worklist->state[1] = 0; worklist->state[2] = 0; sweep.max_reach = 0;
			AlignTask outermost = { 0, 0, target.gene_start, ALIGN_TASK_ROOT | ALIGN_TASK_DELETIONS };
			outermost.flags |= align_iterations(outermost, length, min_score) << ALIGN_TASK_ITERATIONS_SHIFT;
			worklist->push(outermost);
			if (strands > 1) { outermost.flags |= ALIGN_TASK_STRAND; worklist->push(outermost); }
		}
#if defined(__HIP_DEVICE_COMPILE__)
		unsigned long long tick = worklist->stats != nullptr ? wall_clock64() : 0ull;
#define SWEEP_LAP(slot) do { if (worklist->stats != nullptr && lane == 0) { const unsigned long long now_ = wall_clock64(); worklist->stats[slot] += (uint32_t) (now_ - tick); tick = now_; } else if (worklist->stats != nullptr) tick = wall_clock64(); } while (0)
#else
#define SWEEP_LAP(slot) ((void) 0)
#endif
		sync_lanes();
		int32_t looked_up = 0; // read positions whose seeds are known
		for (int32_t block = 0; block + KMER_LENGTH < length; block += (int32_t) ALIGN_SWEEP_BLOCK) {
			if (block >= looked_up) {
				// the seeds of the next 64 read positions (one per lane): the hits of the 8-mer from the start of the gene to its end (source/filter_mismappers.cpp:100-106).  Not of every
				// read position in front of the sweep (until round 6): most sweeps end in the first 64 (max_reach); and the END of the hits is searched from their start in doubling steps --
				// a gene holds a few of the ten thousands of places of an 8-mer
				for (uint32_t item = first_of_mine(width); item < strands * ALIGN_SWEEP_LOOKUPS; item += stride_of_mine(width)) {
					const uint32_t strand = item / ALIGN_SWEEP_LOOKUPS;
					const int32_t read_pos = looked_up + (int32_t) (item % ALIGN_SWEEP_LOOKUPS);
					if (read_pos >= length) continue;
					const Segment read = of_strand(forward, strand);
					uint32_t first = 0, count = 0;
					if (read_pos + KMER_LENGTH < length && target.kmer_offsets != 0) {
						ALIGN_STAT(read_positions, 1);
						const uint32_t kmer = read.kmer((uint32_t) read_pos);
						const uint32_t begin = target.kmer_offsets[kmer], end = target.kmer_offsets[kmer + 1];
						first = lower_bound_i32(target.positions, begin, end, target.gene_start);
						uint32_t behind = first, step = 1; // every hit in [first, behind) lies in front of the end of the gene
						while (behind < end) {
							const uint32_t probe = behind + step - 1 < end - 1 ? behind + step - 1 : end - 1;
							if (target.positions[probe] < target.gene_end) { behind = probe + 1; step <<= 1; }
							else { behind = lower_bound_i32(target.positions, behind, probe, target.gene_end); break; }
						}
						count = behind - first;
					}
					sweep.hit_first[item] = first; sweep.hit_count[item] = count;
				}
				looked_up += (int32_t) ALIGN_SWEEP_LOOKUPS;
				sync_lanes();
				SWEEP_LAP(4);
#if !defined(__HIP_DEVICE_COMPILE__)
				if (round_steps != nullptr) { round_steps[0] += 16 * ((strands * ALIGN_SWEEP_LOOKUPS + width - 1) / width); round_steps[1] += (strands * ALIGN_SWEEP_LOOKUPS + width - 1) / width; } // (the look-ups, in the currency of the steps of a walk)
#endif
			}
			sync_lanes();
			if (worklist->state[1] != 0) return false; // the list of the calls ran over
			const uint32_t listed = worklist->state[0] < worklist->capacity ? worklist->state[0] : worklist->capacity;
			const bool beyond_every_call = (uint32_t) block >= sweep.max_reach; // (a call is listed by a seed at least eight read positions in front of its first one: the calls of this block and all behind it are known)
			sync_lanes(); // (nobody lists a call before everybody has read how many there are)
			if (beyond_every_call) break;
			if (!sweep_collect_calls(sweep, block, listed, width, strands)) { if (lane == 0) worklist->state[1] = 1; sync_lanes(); return false; }
			SWEEP_LAP(5);
#if defined(__HIP_DEVICE_COMPILE__)
			if (worklist->stats != nullptr && lane == 0) worklist->stats[15] += 1;
#endif
			if (sweep.n_calls == 0) continue;
			sync_lanes();
#if defined(__HIP_DEVICE_COMPILE__)
			if (worklist->stats != nullptr && lane == 0) worklist->stats[1] += sweep.n_calls;
#endif
			const uint32_t in_batch = (uint32_t) (block - (looked_up - (int32_t) ALIGN_SWEEP_LOOKUPS)); // the block's place among the read positions looked up last
			if (lane == 0) { // the seeds of the read positions that a call reaches, numbered through, strand by strand
				uint32_t total = 0;
				for (uint32_t slot = 0; slot < strands * ALIGN_SWEEP_BLOCK; ++slot) {
					const uint32_t k = slot % ALIGN_SWEEP_BLOCK;
					if ((sweep.reached >> slot & 1) && block + (int32_t) k < length) total += sweep.hit_count[(slot / ALIGN_SWEEP_BLOCK) * ALIGN_SWEEP_LOOKUPS + in_batch + k];
					sweep.seed_end[slot] = total;
				}
			}
			sync_lanes();
			const uint32_t seeds = sweep.seed_end[strands * ALIGN_SWEEP_BLOCK - 1];
#if defined(__HIP_DEVICE_COMPILE__)
			if (worklist->stats != nullptr && lane == 0) { worklist->stats[2] += seeds; worklist->stats[13] += (seeds + width - 1) / width; worklist->stats[14] += 1; }
			SWEEP_LAP(12);
#endif
			for (uint32_t seed_base = 0; seed_base < seeds; seed_base += width) {
#if !defined(__HIP_DEVICE_COMPILE__)
				unsigned long long longest = 0;
#endif
				for (uint32_t seed = seed_base + first_of_mine(width); seed < seeds && seed < seed_base + width; seed += stride_of_mine(width)) {
					uint32_t slot = 0;
					while (sweep.seed_end[slot] <= seed) ++slot;
					const uint32_t strand = slot / ALIGN_SWEEP_BLOCK, k = slot % ALIGN_SWEEP_BLOCK;
					const int32_t read_pos = block + (int32_t) k;
					const uint32_t hit = sweep.hit_first[strand * ALIGN_SWEEP_LOOKUPS + in_batch + k] + (seed - (slot > 0 ? sweep.seed_end[slot - 1] : 0u));
					const Segment read = of_strand(forward, strand);
#if !defined(__HIP_DEVICE_COMPILE__)
					const unsigned long long before = g_align_seed_steps;
#endif
					if (sweep_seed(sweep, read, target, min_score, read_pos, target.positions[hit], strand)) worklist->state[2] = 1;
#if !defined(__HIP_DEVICE_COMPILE__)
					if (g_align_seed_steps - before > longest) longest = g_align_seed_steps - before;
					if (budget != nullptr) *budget -= (int64_t) (g_align_seed_steps - before); // (host stepping: the steps of the read, for the statistics of the harness)
#endif
				}
#if !defined(__HIP_DEVICE_COMPILE__)
				if (round_steps != nullptr) { round_steps[0] += longest; round_steps[1] += 1; }
#endif
				sync_lanes();
				if (worklist->state[2] != 0) return true; // (the same for every lane: read behind the barrier)
			}
			SWEEP_LAP(6);
		}
		sync_lanes();
#if defined(__HIP_DEVICE_COMPILE__)
		if (worklist->stats != nullptr && lane == 0) worklist->stats[0] += worklist->state[0];
#endif
		return worklist->state[2] != 0;
	}
	// the segment as the search of a strand reads it: forward (base codes in `cache`) or reverse complement (in `cache2`)
	AGPU_HD Segment of_strand(const Segment& forward, uint32_t strand) const {
		Segment read = forward;
		if (strand != 0) { read.reverse_complement = true; read.cache = cache2; read.chars = chars2; }
		return read;
	}
	// both strands of a segment against a gene in one sweep?  (needs what the sweep needs, and room for the base codes of both strands)
	AGPU_HD bool sweeps_strands_together(const Segment& segment, const AlignTarget& target) const {
		return strands_together && worklist != nullptr && worklist->sweep != nullptr && memo != nullptr && cache != nullptr && cache2 != nullptr && segment.length <= cache_capacity
		       && memo->usable(target.gene_start, target.gene_end, (int32_t) segment.length) && segment.length <= ALIGN_SWEEP_SEGMENT;
	}
	// returns whether one of the strands aligns; *decided = false: a list ran over, nothing is known (the caller searches strand by strand)
	AGPU_HD bool align_strands(const Segment& segment, const AlignTarget& target, int32_t min_score, bool* decided) const {
		Segment forward = segment; forward.reverse_complement = false; forward.cache = nullptr; forward.chars = nullptr;
		Segment reverse = segment; reverse.reverse_complement = true; reverse.cache = nullptr; reverse.chars = nullptr;
		sync_lanes(); // (nobody still reads the previous copies)
		for (uint32_t i = lane; i < segment.length; i += lanes) { cache[(size_t) i * cache_stride] = (uint8_t) forward.code(i); cache2[(size_t) i * cache_stride] = (uint8_t) reverse.code(i); }
		if (chars != nullptr && chars2 != nullptr) for (uint32_t i = lane; i < segment.length + 8; i += lanes) { chars[i] = i < segment.length ? (uint8_t) base_char(forward.code(i)) : (uint8_t) 0; chars2[i] = i < segment.length ? (uint8_t) base_char(reverse.code(i)) : (uint8_t) 0; }
		sync_lanes();
		forward.cache = cache; forward.cache_stride = cache_stride; forward.chars = chars != nullptr && chars2 != nullptr ? chars : nullptr;
		new_memo_epoch();
		const bool found = align_by_sweep(forward, 2, target, min_score);
		*decided = worklist->state[1] == 0;
		return found;
	}
	AGPU_HD bool align(const Segment& read, const AlignTarget& target, int32_t min_score) const {
		const int32_t length = (int32_t) read.length;
		if (memo != nullptr) new_memo_epoch(); // a new search: the entries of the previous one (other gene, strand, segment, min_score) must not match
		if constexpr (SWEEP_ONLY) {
			if (memo->usable(target.gene_start, target.gene_end, length) && (uint32_t) length <= ALIGN_SWEEP_SEGMENT) {
				const bool found = align_by_sweep(read, 1, target, min_score);
				if (worklist->state[1] == 0) return found;
			}
			sync_lanes();
			if (lane == 0) *budget = -1; // not a search for the sweep: the read is done again by the kernel that holds the recursion
			sync_lanes();
			return false;
		}
		if (worklist != nullptr && worklist->sweep != nullptr && memo != nullptr && memo->usable(target.gene_start, target.gene_end, length) && (uint32_t) length <= ALIGN_SWEEP_SEGMENT) {
			const bool found = align_by_sweep(read, 1, target, min_score);
			if (worklist->state[1] == 0) return found;
			new_memo_epoch(); // a list was too short for this search: done again by the recursion (what the memo has "seen" are not failures)
		} else if (worklist != nullptr && memo != nullptr && memo->usable(target.gene_start, target.gene_end, length)) {
			sync_lanes();
			if (lane == 0) { worklist->state[0] = 0; worklist->state[1] = 0; worklist->state[2] = 0; } // ([3], the host's consistency flag, is the caller's)
			sync_lanes();
			for (int32_t read_pos = (int32_t) lane; read_pos + KMER_LENGTH < length && 2 * read_pos + min_score <= length + 2 * KMER_LENGTH; read_pos += (int32_t) lanes) { // the iterations of the outermost loop
				const AlignTask outermost = { -read_pos, read_pos, target.gene_start, ALIGN_TASK_ROOT | ALIGN_TASK_DELETIONS };
				worklist->push(outermost);
			}
#if !defined(__HIP_DEVICE_COMPILE__)
			uint32_t tasks_run = 0; // host stepping: every listed task must have been run when the search ends without a success (state[3] = 1 otherwise)
#endif
			for (uint32_t head = 0, taken = 0; ; head += taken) { // rounds: every lane takes one task, the tasks it lists are taken in later rounds
				sync_lanes();
				const uint32_t listed = worklist->state[0] < worklist->capacity ? worklist->state[0] : worklist->capacity;
				const bool done = worklist->state[2] != 0 || head >= listed;
				sync_lanes(); // (nobody lists a task before everybody has read the state of this round)
				if (done) break;
				taken = listed - head < lanes ? listed - head : lanes; // (a round that is not full: the tasks listed during it start behind `listed`, not behind head + lanes)
#if !defined(__HIP_DEVICE_COMPILE__)
				if (virtual_lanes > 1) { // the tasks of one round of the device, one after the other
					taken = listed - head < virtual_lanes ? listed - head : virtual_lanes;
					long long longest = 0;
					tasks_run += taken;
					for (uint32_t v = 0; v < taken; ++v) {
						const long long before = budget != nullptr ? *budget : 0;
						if (align_search(read, target, min_score, stack, 0, worklist->task(head + v), budget, 0, 1, memo, worklist)) worklist->state[2] = 1;
						if (budget != nullptr && before - *budget > longest) longest = before - *budget;
						if (exhausted()) return false;
					}
					if (round_steps != nullptr) { round_steps[0] += (unsigned long long) longest; round_steps[1] += 1; }
					continue;
				}
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
				tasks_run += taken;
#endif
				if (lane < taken && align_search(read, target, min_score, stack, 0, worklist->task(head + lane), budget, 0, 1, memo, worklist)) worklist->state[2] = 1;
				if (exhausted()) return false;
			}
			if (worklist->state[2] != 0) return true;
#if !defined(__HIP_DEVICE_COMPILE__)
			if (worklist->state[1] == 0 && tasks_run != worklist->state[0]) worklist->state[3] = 1;
#endif
			if (worklist->state[1] == 0) return false;
			new_memo_epoch(); // the list was too short for this search: done again by the recursion (what the memo has "seen" are not failures)
		}
		if (lanes_share_seeds) {
			for (int32_t read_pos = 0; read_pos + KMER_LENGTH < length && 2 * read_pos + min_score <= length + 2 * KMER_LENGTH; ++read_pos) {
				const bool found = align_from_read_position(read, target, min_score, stack, max_depth, read_pos, budget, lane, lanes, memo);
				if (exhausted()) return false;
				if (any(found)) return true;
			}
			return false;
		}
		for (int32_t base = 0; base + KMER_LENGTH < length && 2 * base + min_score <= length + 2 * KMER_LENGTH; base += (int32_t) lanes) { // the loop bound of the reference at read_pos = base
			const int32_t read_pos = base + (int32_t) lane;
			bool found = read_pos + KMER_LENGTH < length && 2 * read_pos + min_score <= length + 2 * KMER_LENGTH && align_from_read_position(read, target, min_score, stack, max_depth, read_pos, budget, 0, 1, memo);
			if (exhausted()) return false;
			if (any(found)) return true;
		}
		return false;
	}
};
typedef AlignRunnerT<false> AlignRunner;


// reference: align_both_strands (source/filter_mismappers.cpp:201-245).  `segment` is the part of the read to re-align, read_length the
// length of the whole read; the genes are those of the other end of the fragment.
template <class Runner> AGPU_HD bool align_both_strands(const Segment& segment, int32_t read_length, int32_t max_mate_gap, bool breakpoints_on_same_contig, int32_t alignment_start, int32_t alignment_end,
                                const AnnotationView& ann, const GenomeView& genome, const KmerIndexView& kmers, const SpliceSiteView& splice, const IdSet& genes, float min_align_fraction, const Runner& runner) {
	if (segment.length >= 300) return false; // long reads are not re-aligned
	const int32_t min_score = (int32_t) ((double) (min_align_fraction * (float) segment.length) + 0.5);
	for (uint32_t g = 0; g < genes.n; ++g) {
		const uint32_t gene = genes.get(g);
		const uint32_t contig = ann.gene_contig[gene];
		const int64_t contig_size = (int64_t) (genome.contig_offset[contig + 1] - genome.contig_offset[contig]);
		AlignTarget target;
		target.gene_start = ann.gene_start[gene] - max_mate_gap - read_length; if (target.gene_start < 0) target.gene_start = 0;
		target.gene_end = ann.gene_end[gene] + max_mate_gap + read_length; if ((int64_t) target.gene_end > contig_size - 1) target.gene_end = (int32_t) (contig_size - 1);
		// intragenic events / overlapping genes: donor and acceptor both overlap the breakpoint, the read would always be discarded
		if (breakpoints_on_same_contig && ((alignment_start >= target.gene_start && alignment_start <= target.gene_end) || (alignment_end >= target.gene_start && alignment_end <= target.gene_end)))
			continue;
		const uint32_t table = contig < kmers.n_contigs ? kmers.contig_table[contig] : NO_KMER_TABLE;
		target.kmer_offsets = table == NO_KMER_TABLE ? 0 : kmers.offsets + (size_t) table * KMER_COUNT;
		target.positions = kmers.positions;
		target.contig_bases = genome.bases + genome.contig_offset[contig];
		target.splice_sites = splice.sites + splice.offset[gene]; target.n_splice_sites = splice.offset[gene + 1] - splice.offset[gene];
		target.splice_bits = splice.bits; target.splice_bit_base = genome.contig_offset[contig];
		if (runner.sweeps_strands_together(segment, target)) {
			bool decided = false;
			const bool found = runner.align_strands(segment, target, min_score, &decided);
			if (found) return true;
			if (decided) continue;
		}
		Segment forward = segment; forward.reverse_complement = false;
		if (runner.align(runner.prepared(forward), target, min_score)) return true;
		if (runner.exhausted()) return false;
		Segment reverse = segment; reverse.reverse_complement = true;
		if (runner.align(runner.prepared(reverse), target, min_score)) return true;
		if (runner.exhausted()) return false;
	}
	return false;
}

// reference: extend_split_read (source/filter_mismappers.cpp:247-270): did the aligner clip prematurely?
AGPU_HD bool extend_split_read(const BatchView& b, const GenomeView& genome, uint64_t i, const SequenceRef& sequence, float min_align_fraction) {
	const uint32_t* cigar = cigar_of(b, SPLIT_READ, i); const uint32_t n_cigar = b.cigar_count[SPLIT_READ][i];
	const uint32_t contig = b.contig[SPLIT_READ][i];
	const char* contig_bases = genome.bases + genome.contig_offset[contig];
	const int64_t contig_size = (int64_t) (genome.contig_offset[contig + 1] - genome.contig_offset[contig]);
	int64_t clipped_count, read_from, reference_from;
	if (b.abits[SPLIT_READ][i] & ABIT_STRAND) {
		const int64_t clip = preclipping(cigar, n_cigar), start = b.start[SPLIT_READ][i];
		clipped_count = clip < start ? clip : start; // do not run over the contig boundary
		read_from = clip - clipped_count; reference_from = start - clipped_count;
	} else {
		const int64_t clip = postclipping(cigar, n_cigar), end = b.end[SPLIT_READ][i];
		clipped_count = clip < contig_size - end - 2 ? clip : contig_size - end - 2;
		read_from = (int64_t) sequence.length - clip; reference_from = end + 1;
	}
	if (clipped_count < 0 || read_from < 0) clipped_count = 0; // out of the reference's defined behaviour (substr would throw)
	if (read_from + clipped_count > (int64_t) sequence.length) clipped_count = (int64_t) sequence.length > read_from ? (int64_t) sequence.length - read_from : 0; // substr clamps
	uint32_t matching_bases = 0;
	for (int64_t k = 0; k < clipped_count; ++k) {
		const int64_t reference_position = reference_from + k;
		const char reference_base = (reference_position >= 0 && reference_position < contig_size) ? contig_bases[reference_position] : '\0';
		if (sequence.at((uint32_t) (read_from + k)) == reference_base) ++matching_bases;
	}
	return (float) matching_bases >= floorf((float) clipped_count * min_align_fraction);
}

// Does fragment i support its fusion only because it is mis-mapped?  reference: the per-read part of filter_mismappers
// (source/filter_mismappers.cpp:283-332); a pure function of the fragment, its gene sets, max_mate_gap and the k-mer index.
template <class Runner> AGPU_HD bool is_mismapper(const BatchView& b, const AnnotationView& ann, const GenomeView& genome, const KmerIndexView& kmers, const SpliceSiteView& splice, uint64_t i, int32_t max_mate_gap, const Runner& runner) {
	const float min_align_fraction = 0.8f, min_extended_align_fraction = 0.7f;
	AGPU_IDSET(genes);
	if (b.n_aln[i] == 3) {
		const SequenceRef split_sequence = sequence_of(b, SPLIT_READ, i, no_stage()), mate1_sequence = sequence_of(b, MATE1, i, no_stage());
		const bool same_contig = b.contig[SPLIT_READ][i] == b.contig[SUPPLEMENTARY][i]; // == fusion.contig1 == fusion.contig2
		if (extend_split_read(b, genome, i, split_sequence, min_extended_align_fraction)) return true;
		const uint32_t* split_cigar = cigar_of(b, SPLIT_READ, i); const uint32_t split_n = b.cigar_count[SPLIT_READ][i];
		const uint32_t* mate1_cigar = cigar_of(b, MATE1, i); const uint32_t mate1_n = b.cigar_count[MATE1][i];
		Segment clipped, mate; clipped.sequence = split_sequence; clipped.reverse_complement = false; mate.sequence = mate1_sequence; mate.reverse_complement = false;
		if (b.abits[SPLIT_READ][i] & ABIT_STRAND) {
			uint32_t clip = preclipping(split_cigar, split_n); if (clip > split_sequence.length) clip = split_sequence.length;
			clipped.offset = 0; clipped.length = clip;                                         // sequence.substr(0, preclipping)
			uint32_t mate_clip = preclipping(mate1_cigar, mate1_n); if (mate_clip > mate1_sequence.length) mate_clip = mate1_sequence.length;
			mate.offset = mate_clip; mate.length = mate1_sequence.length - mate_clip;          // mate1.sequence.substr(preclipping)
		} else {
			uint32_t clip = postclipping(split_cigar, split_n); if (clip > split_sequence.length) clip = split_sequence.length;
			clipped.offset = split_sequence.length - clip; clipped.length = clip;             // sequence.substr(length - postclipping)
			uint32_t mate_clip = postclipping(mate1_cigar, mate1_n); if (mate_clip > mate1_sequence.length) mate_clip = mate1_sequence.length;
			mate.offset = 0; mate.length = mate1_sequence.length - mate_clip;                  // mate1.sequence.substr(0, length - postclipping)
		}
		load_genes(b, SPLIT_READ, i, genes);
		if (align_both_strands(clipped, (int32_t) split_sequence.length, max_mate_gap, same_contig, b.start[SUPPLEMENTARY][i], b.end[SUPPLEMENTARY][i], ann, genome, kmers, splice, genes, min_align_fraction, runner))
			return true; // the clipped segment aligns to the donor
		if (runner.exhausted()) return false;
		load_genes(b, SUPPLEMENTARY, i, genes);
		return align_both_strands(mate, (int32_t) mate1_sequence.length, max_mate_gap, same_contig, b.start[MATE1][i], b.end[MATE1][i], ann, genome, kmers, splice, genes, min_align_fraction, runner); // the mate aligns to the acceptor
	}
	// discordant mates: an alignment as long as the chimeric alignment suffices
	const bool same_contig = b.contig[MATE1][i] == b.contig[MATE2][i];
	for (int mate = MATE1; mate <= MATE2; ++mate) {
		const SequenceRef sequence = sequence_of(b, mate, i, no_stage());
		const uint32_t* cigar = cigar_of(b, mate, i); const uint32_t n_cigar = b.cigar_count[mate][i];
		const float clipped_fraction = ((float) preclipping(cigar, n_cigar) + postclipping(cigar, n_cigar)) / sequence.length;
		const float reduced = min_align_fraction * (1 - clipped_fraction);
		Segment whole; whole.sequence = sequence; whole.offset = 0; whole.length = sequence.length; whole.reverse_complement = false;
		load_genes(b, mate == MATE1 ? MATE2 : MATE1, i, genes);
		if (align_both_strands(whole, (int32_t) sequence.length, max_mate_gap, same_contig, b.start[mate][i], b.end[mate][i], ann, genome, kmers, splice, genes, reduced < min_align_fraction ? reduced : min_align_fraction, runner))
			return true;
		if (runner.exhausted()) return false;
	}
	return false;
}

// reference: count_mismappers + the final loop of filter_mismappers (source/filter_mismappers.cpp:247-258, 336-356) for an unfiltered
// candidate: updates its three counters and returns true if the candidate is discarded
AGPU_HD bool count_candidate_mismappers(const BatchView& b, const CandidateTable& t, uint32_t c, float max_mismapper_fraction) {
	const uint64_t* offsets = t.list_offset + 3 * (uint64_t) c;
	uint16_t total_reads = 0, mismappers = 0; // short unsigned int in the reference
	uint32_t* counters[3] = { t.split_reads1 + c, t.split_reads2 + c, t.discordant_mates + c };
	for (int list = 0; list < 3; ++list) {
		uint32_t supporting_reads = *counters[list];
		for (uint64_t k = offsets[list]; k < offsets[list + 1]; ++k) {
			uint8_t filter = b.filter[t.read_lists[k]];
			if (filter == FILTER_none) total_reads++;
			else if (filter == FILTER_mismappers) { total_reads++; mismappers++; if (supporting_reads > 0) supporting_reads--; }
		}
		*counters[list] = supporting_reads;
	}
	return mismappers > 0 && (double) mismappers >= floor((double) (max_mismapper_fraction * (float) total_reads));
}

}

#endif
