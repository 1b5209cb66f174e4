// arriba_amd/csrc/device/agpu_ingest.hip -- read_chimeric_alignments on the MI355X (reference: source/read_chimeric_alignments.cpp:560-773).
//
// The host feeds the bytes of the BAM file; everything else happens in HBM:
//   bgzf_unwrap_kernel        payloads of stored BGZF blocks (STAR --outBAMcompression 0) moved into the contiguous record stream, one workgroup per block
//   segment_guess/check/repair/emit   the record chain (every record starts where the previous one ends) cut in parallel: every 8 KB segment guesses its
//                             first record by the plausibility of two consecutive headers and walks to its end; a guess is accepted only if it is the
//                             exact end of the segment before it, otherwise the segment is walked again from there -- by induction from the known first
//                             record the result is the true chain, whatever the guesses were
//   record_parse_kernel       per record: skip rules, HI / SA aux tags, key of "QNAME,HI" (source/read_chimeric_alignments.cpp:611-631,653)
//   rocPRIM radix sort        records of one name adjacent, file order kept inside (stable); group_head_kernel marks the groups and proves the keys collision-free
//   group_replay_kernel       one thread replays the reference's loop body over the records of one name (ingest_core.hpp): alignment plans, coverage_t, counters;
//                             remove_malformed_alignments on the result
//   name order                fragments ordered by first occurrence; if that is not the order of the names (std::map order, hazard H3) an LSD radix sort
//                             over 8-byte chunks of the names follows
//   fragment_layout/pack      pool offsets by prefix sums in name order, then every fragment writes its columns, CIGARs, 4-bit sequences and name
// All of it is integer / byte work bound by HBM bandwidth and by the latency of dependent loads; no MFMA.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "agpu_context.hpp"
#include "ingest_core.hpp"
#include "device_utils.hpp"
#include "shard_host.hpp"
#include "crc32_core.hpp"
#include "inflate_fast_core.hpp"

using namespace agpu;
namespace agpu { extern thread_local bool g_inside_ingest_finish; } // agpu_api.hip (test hook: agpu_debug_fail_allocation_in_finish)

namespace {

const int BLOCK = 256;
inline unsigned int grid_for(uint64_t n, int block = BLOCK) { return (unsigned int) ((n + block - 1) / block); }

enum { IC_ACTIVE = 0, IC_MAPPED_READS = 1, IC_MISSING_HI = 2, IC_BROKEN = 3, IC_MALFORMED = 4, IC_CHIMERIC = 5, IC_COLLISION = 6, IC_MISMATCH = 7, IC_UNSORTED = 8, IC_MAX_NAME = 9,
       IC_MAX_READ_LENGTH = 10, IC_STRAND_COUNT = 11, IC_STRAND_MATCHING = 12, IC_RUNS_UNSORTED = 13, IC_QNAME_COMMA = 14, IC_COUNT = 16 };

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_NO_MEMORY; } } while (0)
#define TRY(call) do { int s_ = (call); if (s_ != AGPU_OK) return s_; } while (0)

// ---- container ------------------------------------------------------------------------------------------------------------------------------------

// one workgroup per stored block: payload bytes raw -> stream (both sides unaligned; the destination is written in aligned words where possible)
__global__ void __launch_bounds__(BLOCK) bgzf_unwrap_kernel(const uint8_t* raw, const agpu_bgzf_block* blocks, uint8_t* stream) {
	const agpu_bgzf_block block = blocks[blockIdx.x];
	const uint8_t* source = raw + block.raw_offset + block.payload_offset;
	uint8_t* target = stream + block.stream_offset;
	const uint32_t size = block.payload_size;
	uint32_t head = (uint32_t) ((4 - ((uintptr_t) target & 3)) & 3);
	if (head > size) head = size;
	if (threadIdx.x < head) target[threadIdx.x] = source[threadIdx.x];
	const uint32_t words = (size - head) / 4;
	for (uint32_t w = threadIdx.x; w < words; w += BLOCK) *(uint32_t*) (target + head + 4 * (size_t) w) = load_u32(source + head + 4 * (size_t) w);
	const uint32_t tail = head + 4 * words;
	if (threadIdx.x < size - tail) target[tail + threadIdx.x] = source[tail + threadIdx.x];
}

// Deflated blocks (round 5: inflate_fast_core.hpp).  Pass 1: a LANE per block, INFLATE_LANES blocks per workgroup (one wavefront whose lanes run the same loop on different
// blocks; the tables of a block are 2 KB of LDS): literals to their place, matches noted.  The first / last block of a part of a file gives only the bytes [skip, skip + keep) of
// what it holds: such a block is inflated into `spill` (64 KB for block 0, 64 KB for the last one) and its share copied from there by pass 2.
template <int LANES> __global__ void __launch_bounds__(LANES) bgzf_inflate_tokens_kernel(const uint8_t* raw, const agpu_bgzf_block* blocks, uint32_t n_blocks, uint8_t* stream, uint8_t* spill,
                                                                                        unsigned long long* notes, uint32_t* note_count, int* status, unsigned int* failures) {
	__shared__ InflateFastTables tables[LANES];
	const uint32_t b = blockIdx.x * LANES + threadIdx.x;
	if (b >= n_blocks) return;
	const agpu_bgzf_block block = blocks[b];
	const bool partial = block.skip != 0 || block.keep != block.isize;
	uint8_t* target = partial ? spill + (b == 0 ? 0 : 65536) : stream + block.stream_offset;
	uint32_t noted = 0;
	const int result = block.isize > 65536 ? (int) INFLATE_OUTPUT_OVERRUN
	                 : inflate_tokens(raw + block.raw_offset + block.payload_offset, block.payload_size, target, block.isize, notes + (size_t) b * INFLATE_MATCH_CAPACITY, INFLATE_MATCH_CAPACITY, noted, tables[threadIdx.x]);
	note_count[b] = noted; status[b] = result;
	if (result != INFLATE_OK && result != INFLATE_RETRY) atomicAdd(failures, 1u);
}

// Pass 2: a wavefront per block, the noted matches 64 at a time, a lane per match.  A match is copied as soon as the bytes it reads are final (inflate_match_is_ready): all
// bytes in front of the first match that is still pending are.  The first pending match is always ready, so every round copies at least one; a group whose matches read from in
// front of the group (most) is done in one round.
__global__ void __launch_bounds__(BLOCK) bgzf_inflate_resolve_kernel(const agpu_bgzf_block* blocks, uint32_t n_blocks, uint8_t* stream, uint8_t* spill, const unsigned long long* notes, const uint32_t* note_count, const int* status) {
	const uint32_t b = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (b >= n_blocks || status[b] != INFLATE_OK) return;
	const agpu_bgzf_block block = blocks[b];
	const bool partial = block.skip != 0 || block.keep != block.isize;
	uint8_t* target = partial ? spill + (b == 0 ? 0 : 65536) : stream + block.stream_offset;
	const unsigned long long* mine = notes + (size_t) b * INFLATE_MATCH_CAPACITY;
	const uint32_t n = note_count[b];
	for (uint32_t base = 0; base < n; base += 64) {
		const unsigned long long note = base + lane < n ? mine[base + lane] : 0ull;
		const uint32_t position = inflate_note_position(note), length = inflate_note_length(note), distance = inflate_note_distance(note);
		bool pending = base + lane < n;
		while (true) {
			const unsigned long long waiting = __ballot(pending);
			if (waiting == 0) break;
			const uint32_t frontier = (uint32_t) __shfl((int) position, __ffsll((unsigned long long) waiting) - 1);
			if (pending && inflate_match_is_ready(position, length, distance, frontier)) { inflate_copy_match(target, position, length, distance); pending = false; }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // what this round wrote is what the next round reads
		}
	}
	if (partial) for (uint32_t k = lane; k < block.keep; k += 64) stream[block.stream_offset + k] = target[block.skip + k];
}

// The decoder of round 4: one wavefront per deflated block (inflate_core.hpp).  It takes the blocks pass 1 hands back (INFLATE_RETRY: more matches than it has room to note)
// -- status == nullptr: all blocks (ARRIBA_INFLATE=wave, the measured alternative).
__global__ void __launch_bounds__(64, 3) bgzf_inflate_kernel(const uint8_t* raw, const agpu_bgzf_block* blocks, uint8_t* stream, uint8_t* spill, int* status, unsigned int* failures) {
	__shared__ InflateShared shared;
	if (status != nullptr && status[blockIdx.x] != INFLATE_RETRY) return;
	const agpu_bgzf_block block = blocks[blockIdx.x];
	const bool partial = block.skip != 0 || block.keep != block.isize;
	uint8_t* target = partial ? spill + (blockIdx.x == 0 ? 0 : 65536) : stream + block.stream_offset;
	auto sync = [] () { __syncthreads(); };
	auto broadcast = [] (uint32_t value) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) value); };
	const int result = inflate_block(raw + block.raw_offset + block.payload_offset, block.payload_size, target, block.isize, shared, threadIdx.x, 64u, sync, broadcast);
	if (result != INFLATE_OK) { if (threadIdx.x == 0) atomicAdd(failures, 1u); return; }
	if (partial) {
		__syncthreads();
		for (uint32_t k = threadIdx.x; k < block.keep; k += 64) stream[block.stream_offset + k] = __hip_atomic_load(&target[block.skip + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

// ---- the record chain -----------------------------------------------------------------------------------------------------------------------------

// (all four over the segments [segment_begin, n_segments): the whole stream at once, or the segments of one window of it behind those of the windows before)
__global__ void segment_guess_kernel(const uint8_t* bytes, uint64_t size, uint64_t base, uint64_t segment_begin, uint64_t n_segments, uint32_t n_targets, uint64_t* first, uint64_t* end, uint32_t* count) {
	const uint64_t s = segment_begin + blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (s < n_segments) guess_segment(bytes, size, base, s, n_targets, first, end, count);
}

__global__ void segment_check_kernel(const uint64_t* first, const uint64_t* end, uint64_t segment_begin, uint64_t n_segments, uint8_t* mismatch, uint32_t* mismatches) {
	__shared__ uint32_t block_sum;
	uint32_t mine = 0;
	for (uint64_t s = segment_begin + blockIdx.x * (uint64_t) BLOCK + threadIdx.x; s < n_segments; s += gridDim.x * (uint64_t) BLOCK) {
		const bool bad = s > 0 && first[s] != end[s - 1];
		mismatch[s] = bad;
		mine += bad;
	}
	block_tally(mine, mismatches, &block_sum);
}

__global__ void segment_repair_kernel(const uint8_t* bytes, uint64_t size, uint64_t base, uint64_t segment_begin, uint64_t n_segments, const uint8_t* mismatch, const uint64_t* previous_end, uint64_t* first, uint64_t* end, uint32_t* count) {
	const uint64_t s = segment_begin + blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (s < n_segments) repair_segment_run(bytes, size, base, n_segments, s, mismatch, previous_end, first, end, count);
}

__global__ void segment_emit_kernel(const uint8_t* bytes, uint64_t size, uint64_t base, uint64_t segment_begin, uint64_t n_segments, const uint64_t* first, const uint32_t* record_base, uint64_t* record_offset) {
	const uint64_t s = segment_begin + blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (s >= n_segments) return;
	const uint64_t begin = base + s * SEGMENT_BYTES, segment_end = (begin + SEGMENT_BYTES < size) ? begin + SEGMENT_BYTES : size;
	uint32_t n = 0;
	walk_segment(bytes, size, first[s], segment_end, n, record_offset + record_base[s]);
}

// ---- per record -----------------------------------------------------------------------------------------------------------------------------------

// Round 5: the 64 records of a wavefront lie next to each other in the stream (12.9 KB at 202 bytes per record), and a lane reads its record's header, name and tags -- a third of
// its bytes, in a dozen loads whose lines the 10^4 wavefronts in flight push out of the L2 between one load and the next: the counters saw the stream fetched 5.4 times over
// (profiles/r05t_pmc100m_pmc_summary.txt).  So the wavefront copies its piece of the stream into LDS first, 16 bytes per lane and turn, every line once, and the lanes parse there.
// A piece that does not fit (reads of some kilobases) is parsed from the stream as before.
const uint32_t PARSE_WINDOW = 16384;
__global__ void __launch_bounds__(BLOCK) record_parse_kernel(IngestStream in, GenomeView genome, uint64_t seed, uint64_t record_begin, uint64_t* keys, uint8_t* bits, int32_t* hit_index, uint32_t* counters) {
	__shared__ uint32_t sums[4];
	__shared__ __attribute__((aligned(16))) uint8_t staged[BLOCK / 64][PARSE_WINDOW + 64];
	const uint32_t wave = threadIdx.x / 64, lane = threadIdx.x % 64;
	uint8_t* window = staged[wave];
	uint32_t active = 0, mapped = 0, missing = 0, broken = 0;
	for (uint64_t base = record_begin + ((uint64_t) blockIdx.x * (BLOCK / 64) + wave) * 64; base < in.n_records; base += (uint64_t) gridDim.x * BLOCK) { // (the same for the lanes of a wavefront)
		const uint64_t r = base + lane;
		const uint64_t piece_begin = in.record_offset[base] & ~15ull, piece_end = base + 64 < in.n_records ? in.record_offset[base + 64] : in.size; // (every record ends where the next one begins)
		const bool in_lds = piece_end > piece_begin && piece_end - piece_begin <= PARSE_WINDOW;
		if (in_lds) {
			for (uint64_t at = piece_begin + 16 * lane; at < piece_end; at += 16 * 64) *(uint4*) (window + (at - piece_begin)) = *(const uint4*) (in.bytes + at); // (the stream is padded behind its end)
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		}
		if (r < in.n_records) {
			const uint64_t offset = in.record_offset[r];
			const uint8_t* p = in.bytes + offset;
			uint32_t block_size = load_u32(p);
			if (in_lds && offset + 4 + (uint64_t) block_size <= piece_end && offset + 36 <= piece_end) { p = window + (offset - piece_begin); block_size = load_u32(p); } // (a record that claims more than its piece holds: from the stream)
			uint64_t key = ~0ull;
			uint8_t status = RECORD_SKIPPED;
			int32_t hit = HIT_INDEX_UNKNOWN;
			if (!record_sizes_ok(p + 4, block_size)) { status = RECORD_BROKEN; ++broken; }
			else {
				const Rec record = load_record_at(in, p);
				if (!((record.flag & BAMF_UNMAP) || ((record.flag & BAMF_PAIRED) && (record.flag & BAMF_MUNMAP)))) {
					const AuxTags tags = scan_aux(record.aux, record.end);
					if (!tags.has_hi && (record.flag & BAMF_SECONDARY)) { status = RECORD_MISSING_HI; ++missing; }
					else if (record.contig < 0) { status = RECORD_BROKEN; ++broken; } // reference id outside the header
					else {
						status = RECORD_ACTIVE | (tags.has_sa ? RECORD_HAS_SA : 0);
						key = name_key(record, tags.has_hi ? tags.hi : 1, seed);
						hit = hit_index_to_keep(tags);
						++active;
						if (!(record.flag & BAMF_SUPPLEMENTARY) && (genome.contig_bits[record.contig] & CBIT_INTERESTING)) ++mapped;
					}
				}
			}
			keys[r] = key;
			bits[r] = status;
			hit_index[r] = hit;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); // (the window is overwritten by the next turn)
	}
	block_tally(active, &counters[IC_ACTIVE], &sums[0]);
	block_tally(mapped, &counters[IC_MAPPED_READS], &sums[1]);
	block_tally(missing, &counters[IC_MISSING_HI], &sums[2]);
	block_tally(broken, &counters[IC_BROKEN], &sums[3]);
}

// position p of the sorted records starts a group if its key differs from the key before it
__global__ void group_head_kernel(const uint64_t* sorted_keys, uint64_t n_active, uint8_t* head) {
	const uint64_t p = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (p < n_active) head[p] = p == 0 || sorted_keys[p] != sorted_keys[p - 1];
}
// The groups are found in the order of their keys (hashes of the names), which scatters them over the stream: a wavefront that replays 64 neighbouring groups of that
// order touches 64 x 3 places of a 54 GB stream.  In the order of their FIRST RECORDS neighbouring groups are neighbours in the stream (STAR writes the alignments of a read
// next to each other), so the loads of a wavefront fall into a few consecutive kilobytes.  The rank of a group in that order = the number of first records in front of its own:
// a flag per record, a prefix sum over the records, one scatter.
__global__ void group_first_flag_kernel(const uint32_t* sorted_records, const uint32_t* group_start, uint32_t n_groups, uint8_t* first_flags) {
	const uint32_t g = blockIdx.x * BLOCK + threadIdx.x;
	if (g < n_groups) first_flags[sorted_records[group_start[g]]] = 1;
}
__global__ void group_rank_kernel(const uint32_t* sorted_records, const uint32_t* group_start, uint32_t n_groups, uint64_t n_active, const uint32_t* stream_rank, uint32_t* group_first, uint32_t* group_begin, uint32_t* group_count) {
	const uint32_t g = blockIdx.x * BLOCK + threadIdx.x;
	if (g >= n_groups) return;
	const uint32_t begin = group_start[g], first = sorted_records[begin], t = stream_rank[first];
	group_first[t] = first; group_begin[t] = begin; group_count[t] = (uint32_t) ((g + 1 < n_groups ? (uint64_t) group_start[g + 1] : n_active) - begin);
}
// equal keys must be equal names: every record of a group against the first one (in the order of the stream)
__global__ void group_names_kernel(IngestStream in, const uint32_t* sorted_records, const uint32_t* group_begin, const uint32_t* group_count, uint32_t first_group, uint32_t n_groups, uint32_t* counters) {
	const uint32_t t = first_group + blockIdx.x * BLOCK + threadIdx.x;
	if (t >= n_groups || group_count[t] < 2) return;
	const uint32_t* records = sorted_records + group_begin[t];
	const Rec a = load_record(in, records[0]);
	const int64_t hit_a = hit_index_of(in, records[0], a);
	for (uint32_t k = 1; k < group_count[t]; ++k) {
		const Rec b = load_record(in, records[k]);
		if (!same_name(a, hit_a, b, hit_index_of(in, records[k], b))) { atomicOr(&counters[IC_COLLISION], 1u); return; }
	}
}

// ---- the front as the pieces arrive: small kernels of the windows -----------------------------------------------------------------------------------

// the record numbers of a window start behind those of the windows before it: their total, kept on the device, is added to the window's own prefix sums ...
__global__ void window_carry_kernel(uint32_t* record_base, uint64_t segment_begin, uint64_t segment_end, const uint32_t* records_before) {
	const uint64_t s = segment_begin + blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (s <= segment_end) record_base[s] += *records_before;
}
// ... and becomes the total behind this window; words: see INGEST_WINDOW_WORDS
__global__ void window_summary_kernel(const uint32_t* record_base, const uint64_t* end, uint64_t segment_end, uint32_t* records_before, uint32_t* words) {
	*records_before = record_base[segment_end];
	words[1] = record_base[segment_end];
	const uint64_t last_end = end[segment_end - 1];
	words[4] = (uint32_t) last_end; words[5] = (uint32_t) (last_end >> 32);
}
struct RecordIsActive { __host__ __device__ bool operator()(uint8_t bits) const { return (bits & RECORD_STATUS_MASK) == RECORD_ACTIVE; } };
// STAR writes the alignments of a read next to each other: the active records in the order of the stream, a new run wherever the key of the name changes.  (Whether a name
// has a second run somewhere else is checked at the end, over the keys of all runs: run_repeat_kernel.)
__global__ void run_head_kernel(const uint64_t* keys, const uint32_t* active_records, uint64_t begin, uint64_t end, uint8_t* head) {
	const uint64_t i = begin + blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i < end) head[i] = i == 0 || keys[active_records[i]] != keys[active_records[i - 1]];
}
// the runs [first_group, n_groups) are complete: their first records and sizes (the run behind the last one starts at run_begin[g + 1], or the active records end at active_total)
__global__ void run_close_kernel(const uint32_t* active_records, const uint32_t* run_begin, uint32_t first_group, uint32_t n_groups, uint32_t n_runs, uint64_t active_total, uint32_t* group_first, uint32_t* group_count) {
	const uint32_t g = first_group + blockIdx.x * BLOCK + threadIdx.x;
	if (g >= n_groups) return;
	const uint32_t begin = run_begin[g];
	group_first[g] = active_records[begin];
	group_count[g] = (uint32_t) ((g + 1 < n_runs ? (uint64_t) run_begin[g + 1] : active_total) - begin);
}
__global__ void run_key_kernel(const uint64_t* keys, const uint32_t* group_first, uint32_t n_groups, uint64_t* out) {
	const uint32_t g = blockIdx.x * BLOCK + threadIdx.x;
	if (g < n_groups) out[g] = keys[group_first[g]];
}
__global__ void run_repeat_kernel(const uint64_t* sorted_keys, uint32_t n_groups, uint32_t* repeats) {
	__shared__ uint32_t block_sum;
	uint32_t mine = 0;
	for (uint64_t g = blockIdx.x * (uint64_t) BLOCK + threadIdx.x; g < n_groups; g += gridDim.x * (uint64_t) BLOCK) mine += g > 0 && sorted_keys[g] == sorted_keys[g - 1];
	block_tally(mine, repeats, &block_sum);
}

struct ViralCounter {
	unsigned long long* counts;
	__device__ void operator()(uint32_t contig) { atomicAdd(&counts[contig], 1ull); }
};

// (Tried in round 3 and taken back: one wavefront per 64 groups with its slice of the stream copied into 48 KB of LDS -- 1240 ms instead of 455 ms at 10^8 fragments,
// profiles/r03j_bench100m_lds_staged_replay.json: three wavefronts per CU cannot hide the look-ups that stay in HBM -- offsets, record bits, gene index, genome, coverage_t.)
// RUNS: the groups are the runs of the windows (neighbouring active records with one key), in the order of the stream.  Then the thread also does what two passes of their own
// over the same records did before (group_names_kernel 149 ms + run_name_order_kernel 123 ms of kernel time at 10^8 fragments, profiles/r04w_serialized_kernels.json): equal keys
// must be equal names -- every record against the first one inside the loop body, while its header is at hand --, and the order of the names of neighbouring runs:
//   Is the name of this run's fragment greater than the name of the run before it?  If so for all runs -- and the "ITD" entry of a run, where it has a valid one, is smaller than
//   the next name, too: run_itd_order_kernel behind this one, which knows the valid entries -- the valid fragments are in name order (the names of the runs between two of them
//   chain), and the pass over the fragments behind the last piece is not needed; if not -- names in FASTQ order, as STAR writes them, or a run out of order -- that pass decides.
//   And for the read-name groups of the batch (fragments with one QNAME: multi-mappers): does the QNAME of this run differ from the one before?  With the runs in name order and
//   no comma inside a QNAME, the runs of one QNAME lie next to each other, so two fragments have one QNAME exactly if no run between them differs from its predecessor ("X,a" <
//   "Y,b" < "X,c" makes "Y,b" start with "X,": Y would hold a comma) -- a prefix sum over these flags then stands in for the QNAMEs (fragment_layout_kernel).
// Registers: left alone the compiler takes 100 VGPRs for this kernel = 4 wavefronts per SIMD.  Held to 64 VGPRs it runs 8 per SIMD for 96 more bytes of scratch per lane
// (-Rpass-analysis=kernel-resource-usage) -- and is SLOWER: 765 instead of 546 ms of kernel time at 10^8 fragments (profiles/r04v_serialized_kernels_fused_replay*.log; the pack
// 130 instead of 124 ms): what the lanes of a CU have open of the stream at once no longer fits its share of the L2, and every line comes from further away again for the next
// field of the record.  AGPU_REPLAY_WAVES / AGPU_PACK_WAVES = n (make variant) holds the kernels to n wavefronts per SIMD, for such comparisons; 0 = the compiler's choice.
#ifndef AGPU_REPLAY_WAVES
#define AGPU_REPLAY_WAVES 0
#endif
#ifndef AGPU_PACK_WAVES
#define AGPU_PACK_WAVES 0
#endif
#if AGPU_REPLAY_WAVES > 0
#define AGPU_REPLAY_OCCUPANCY __attribute__((amdgpu_waves_per_eu(AGPU_REPLAY_WAVES, AGPU_REPLAY_WAVES)))
#else
#define AGPU_REPLAY_OCCUPANCY
#endif
#if AGPU_PACK_WAVES > 0
#define AGPU_PACK_OCCUPANCY __attribute__((amdgpu_waves_per_eu(AGPU_PACK_WAVES, AGPU_PACK_WAVES)))
#else
#define AGPU_PACK_OCCUPANCY
#endif
template <bool RUNS> __global__ void __launch_bounds__(BLOCK) AGPU_REPLAY_OCCUPANCY group_replay_kernel(IngestContext ctx, const uint32_t* sorted_records, const uint32_t* group_begin, const uint32_t* group_count, uint32_t first_group, uint32_t n_groups,
                                                             FragmentPlan* plain, TandemPlan* itd, uint8_t* valid, FragmentSizes* sizes, unsigned long long* viral_counts, uint32_t* qname_differs, uint32_t* counters) {
	__shared__ uint32_t sums[2];
	GroupTally tally; tally.malformed = 0; tally.chimeric = 0; tally.collision = 0;
	const uint32_t g = first_group + blockIdx.x * BLOCK + threadIdx.x; // (groups numbered in the order of their first records)
	if (g < n_groups) {
		const uint32_t begin = group_begin[g];
		const uint32_t n_records = group_count[g];
		const Rec representative = load_record(ctx.stream, sorted_records[begin]);
		const int64_t hit = hit_index_of(ctx.stream, sorted_records[begin], representative);
		const uint32_t length = qname_length(representative);
		if (RUNS) {
			if (find_byte(representative.name, length, ',') < length) atomicOr(&counters[IC_QNAME_COMMA], 1u);
			if (g == 0) qname_differs[0] = 0;
			else {
				const uint32_t record_before = sorted_records[group_begin[g - 1]];
				const Rec run_before = load_record(ctx.stream, record_before);
				const uint32_t length_before = qname_length(run_before), shorter = length < length_before ? length : length_before, different = first_difference(run_before.name, representative.name, shorter);
				qname_differs[g] = length != length_before || different < shorter;
				bool unsorted;
				if (different < shorter) unsorted = run_before.name[different] > representative.name[different]; // (the names differ inside their QNAMEs: that byte decides)
				else unsorted = compare_names(fragment_name(run_before, hit_index_of(ctx.stream, record_before, run_before), false, length_before), fragment_name(representative, hit, false, length)) >= 0;
				if (unsorted) atomicOr(&counters[IC_RUNS_UNSORTED], 1u);
			}
		}
		FragmentPlan plain_plan; TandemPlan itd_plan;
		ViralCounter viral = { viral_counts };
		replay_group(ctx, sorted_records + begin, n_records, plain_plan, itd_plan, tally, viral, RUNS ? &representative : nullptr, length, hit);
		if (tally.collision) atomicOr(&counters[IC_COLLISION], 1u);
		Fragment3 fragment;
		AGPU_NOUNROLL for (uint32_t entry = 0; entry < 2; ++entry) { // the fragment of the name, then its "ITD" entry (one copy of the sanity check in the code: the kernel is 140 KB of instructions as it is)
			const FragmentPlan& plan = entry ? itd_plan.plan : plain_plan;
			bool ok = false;
			if (plan.count > 0) {
				ok = normalize_plan(ctx.stream, plan, entry ? &itd_plan.tandem : nullptr, fragment);
				if (!ok) tally.malformed++;
			}
			valid[2 * (size_t) g + entry] = ok;
			FragmentSizes mine; mine.cigar_words = 0; mine.sequence_bytes = 0; mine.name_length = 0;
			if (ok) { if (entry) itd[g] = itd_plan; else plain[g] = plain_plan; fragment_sizes(fragment, representative, hit, entry, mine, length); }
			sizes[2 * (size_t) g + entry] = mine;
		}
	}
	block_tally(tally.malformed, &counters[IC_MALFORMED], &sums[0]);
	block_tally(tally.chimeric, &counters[IC_CHIMERIC], &sums[1]);
}

// ---- name order -----------------------------------------------------------------------------------------------------------------------------------

// fragment reference = 2 * group + (1 for the "ITD" entry); the groups are numbered by their first records, so ascending references are the fragments in the order of
// their first occurrence, the ITD entry behind the plain one
__device__ FragmentName name_of(const IngestStream& in, const uint32_t* group_first, uint32_t ref, Rec& storage) {
	storage = load_record(in, group_first[ref >> 1]);
	return fragment_name(storage, hit_index_of(in, group_first[ref >> 1], storage), ref & 1);
}

__global__ void name_order_check_kernel(IngestStream in, const uint32_t* group_first, const uint32_t* order, uint64_t n, uint32_t* counters) {
	__shared__ uint32_t block_max;
	if (threadIdx.x == 0) block_max = 0;
	__syncthreads();
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i < n) {
		Rec storage_a, storage_b;
		const FragmentName mine = name_of(in, group_first, order[i], storage_a);
		atomicMax(&block_max, name_length(mine));
		if (i > 0) {
			const FragmentName before = name_of(in, group_first, order[i - 1], storage_b);
			if (compare_names(before, mine) >= 0) atomicOr(&counters[IC_UNSORTED], 1u);
		}
	}
	__syncthreads();
	if (threadIdx.x == 0 && block_max) atomicMax(&counters[IC_MAX_NAME], block_max);
}

// ... the second half of the order of the runs (group_replay_kernel<true> has the first): a run with a valid "ITD" entry -- few have one -- must have that name, "QNAME,HIITD", in
// front of the name of the next run, too
__global__ void run_itd_order_kernel(IngestStream in, const uint32_t* group_first, const uint8_t* valid, uint32_t first_group, uint32_t n_groups, uint32_t* counters) {
	const uint32_t g = first_group + blockIdx.x * BLOCK + threadIdx.x;
	if (g >= n_groups || g == 0 || !valid[2 * (size_t) (g - 1) + 1]) return;
	Rec storage_a, storage_b;
	const FragmentName mine = name_of(in, group_first, 2 * g, storage_b);
	const FragmentName before = name_of(in, group_first, 2 * (g - 1) + 1, storage_a);
	if (compare_names(before, mine) >= 0) atomicOr(&counters[IC_RUNS_UNSORTED], 1u);
}

__global__ void name_chunk_kernel(IngestStream in, const uint32_t* group_first, const uint32_t* order, uint64_t n, uint32_t chunk, uint64_t* keys) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= n) return;
	Rec storage;
	keys[i] = name_chunk(name_of(in, group_first, order[i], storage), chunk);
}

// ---- layout and pack ------------------------------------------------------------------------------------------------------------------------------

__global__ void fragment_layout_kernel(IngestStream in, const uint32_t* group_first, const uint32_t* order, uint64_t n, const FragmentSizes* sizes, const uint32_t* qname_run /* may be null */,
                                       uint32_t* cigar_words, uint32_t* sequence_bytes, uint32_t* name_lengths, uint32_t* new_group) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i > n) return;
	if (i == n) { cigar_words[i] = 0; sequence_bytes[i] = 0; name_lengths[i] = 0; return; } // the element behind the last one carries the totals of the exclusive scans
	const uint32_t ref = order[i];
	const FragmentSizes mine = sizes[ref];
	cigar_words[i] = mine.cigar_words; sequence_bytes[i] = mine.sequence_bytes; name_lengths[i] = mine.name_length;
	// multi-mapper groups: identical names up to the last ',' (source/common.hpp:222), i.e. identical QNAMEs
	uint32_t differs = 0;
	if (i > 0 && qname_run != nullptr) differs = qname_run[ref >> 1] != qname_run[order[i - 1] >> 1]; // (told by the windows: group_replay_kernel<true>, run_itd_order_kernel)
	else if (i > 0) {
		const Rec a = load_record(in, group_first[ref >> 1]), b = load_record(in, group_first[order[i - 1] >> 1]);
		const uint32_t length = qname_length(a);
		differs = length != qname_length(b);
		for (uint32_t k = 0; k < length && !differs; ++k) differs = a.name[k] != b.name[k];
	}
	new_group[i] = differs;
}

__global__ void __launch_bounds__(BLOCK) AGPU_PACK_OCCUPANCY fragment_pack_kernel(IngestStream in, const uint32_t* group_first, const uint32_t* order, uint64_t n, const FragmentPlan* plain, const TandemPlan* itd,
                                                              const uint64_t* cigar_base, const uint64_t* sequence_base, const uint64_t* name_base, const uint32_t* group_id, PackTarget out, uint32_t* counters) {
	__shared__ uint32_t block_max;
	if (threadIdx.x == 0) block_max = 0;
	__syncthreads();
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i == n) out.name_offset[n] = name_base[n];
	if (i < n) {
		const uint32_t ref = order[i], g = ref >> 1;
		const bool is_itd = ref & 1;
		Fragment3 f;
		normalize_plan(in, is_itd ? itd[g].plan : plain[g], is_itd ? &itd[g].tandem : nullptr, f);
		Rec storage;
		const FragmentName name = name_of(in, group_first, ref, storage);
		atomicMax(&block_max, write_fragment(in, f, name, i, cigar_base[i], sequence_base[i], name_base[i], group_id[i], out));
	}
	__syncthreads();
	if (threadIdx.x == 0 && block_max) atomicMax(&counters[IC_MAX_READ_LENGTH], block_max);
}

// CRC-32 of the payload of every stored block of a pushed piece against the trailer of the block (crc32_core.hpp; htslib checks every block it reads: bgzf.c).  Round 5: one
// WAVEFRONT per block (four blocks per workgroup, which share the tables in LDS): lane L takes chunk L of the virtual 64 KB block that ends with the payload, 16 words per round
// through LDS (read from HBM once, sixteen neighbouring lanes 64 consecutive bytes; chunk c at word 17 c -- an odd stride, so that the lanes, each on its own chunk, hit
// different banks), four bytes per step with the sliced tables; the 64 chunk CRCs are folded with two table-driven operators (crc32_core.hpp: Crc32Tables::join) in 14 steps.
// Blocks of which only a part is delivered (the ends of a part of a file) carry crc32 = 0 and are not checked.
// INFLATED: the blocks were deflated and `raw` is the stream they were inflated into: the CRC-32 of the gzip trailer is that of the inflated bytes
template <bool INFLATED> __global__ void __launch_bounds__(256) bgzf_crc_kernel(const uint8_t* raw, const agpu_bgzf_block* blocks, uint32_t n_blocks, const Crc32Tables* tables, unsigned int* mismatches) {
	__shared__ uint32_t slice[4][256];
	__shared__ uint32_t join[2][4][256];
	__shared__ uint32_t staged[4][CRC32_WAVE_LANES * (CRC32_ROUND_WORDS + 1)];
	for (uint32_t k = threadIdx.x; k < 4 * 256; k += 256) ((uint32_t*) slice)[k] = ((const uint32_t*) tables->slice)[k];
	for (uint32_t k = threadIdx.x; k < 2 * 4 * 256; k += 256) ((uint32_t*) join)[k] = ((const uint32_t*) tables->join)[k];
	const uint32_t wave = threadIdx.x / CRC32_WAVE_LANES, lane = threadIdx.x % CRC32_WAVE_LANES, b = blockIdx.x * 4 + wave;
	agpu_bgzf_block block;
	block.crc32 = 0; block.payload_size = 0; block.raw_offset = 0; block.payload_offset = 0;
	if (b < n_blocks) block = blocks[b];
	if (INFLATED) { block.raw_offset = block.stream_offset; block.payload_offset = 0; block.payload_size = block.isize; }
	const bool checked = b < n_blocks && block.crc32 != 0 && block.payload_size <= CRC32_VIRTUAL; // (the same for the lanes of a wavefront; a BGZF block holds at most 64 KB)
	const uint8_t* payload = raw + block.raw_offset + block.payload_offset;
	const uint32_t n = block.payload_size;
	uint32_t* mine = staged[wave];
	uint32_t c = 0;
	const bool by_chunks = checked && n >= 4;
	const uint32_t first_word = by_chunks ? (CRC32_VIRTUAL - n) / 4 : CRC32_VIRTUAL / 4; // the first word of the virtual block that holds a byte of the payload
	for (uint32_t round = 0; round < CRC32_WAVE_CHUNK / 4 / CRC32_ROUND_WORDS; ++round) {
		__syncthreads(); // (the tables are there; the words of the round before have been used)
		for (uint32_t item = lane; item < CRC32_WAVE_LANES * CRC32_ROUND_WORDS; item += CRC32_WAVE_LANES) {
			const uint32_t chunk = item / CRC32_ROUND_WORDS, j = item % CRC32_ROUND_WORDS, w = chunk * (CRC32_WAVE_CHUNK / 4) + round * CRC32_ROUND_WORDS + j;
			mine[chunk * (CRC32_ROUND_WORDS + 1) + j] = w >= first_word ? crc32_virtual_word(payload, n, w) : 0u;
		}
		__syncthreads();
		for (uint32_t j = 0; j < CRC32_ROUND_WORDS; ++j) c = crc32_raw_step(slice, c, mine[lane * (CRC32_ROUND_WORDS + 1) + j]);
	}
	// the chunks of a group of eight, then the eight groups (crc32_fold_chunks)
	__syncthreads();
	mine[lane] = c;
	__syncthreads();
	if (lane % 8 == 0) { for (uint32_t i = 1; i < 8; ++i) c = crc32_apply(join[0], c) ^ mine[lane + i]; }
	__syncthreads();
	if (lane % 8 == 0) mine[lane] = c;
	__syncthreads();
	if (lane == 0 && checked) {
		for (uint32_t g = 1; g < 8; ++g) c = crc32_apply(join[1], c) ^ mine[8 * g];
		c ^= 0xFFFFFFFFu;
		if (n < 4) c = crc32_of_sliced(slice, payload, n); // (no four bytes to invert: the few bytes as they are)
		if (c != block.crc32) atomicAdd(mismatches, 1u);
	}
}

// agpu_shard_merge: a 32-bit column of one part into its place in the whole, pool offsets moved behind the pools of the parts before it
// (slot < 3: only where the fragment has that alignment -- the unused slots of a row hold 0, as fragment_pack_kernel leaves them)
__global__ void shard_rebase_kernel(uint32_t* out, const uint32_t* in, uint64_t n, uint32_t base, const uint8_t* n_aln, uint32_t slot) {
	const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = in[i] + ((slot >= 3 || slot < n_aln[i]) ? base : 0u);
}
// ... the coverage of one part added to the whole: windows before their saturation (sums), start / end flags (ORs), viral read counts (sums)
// ... the name offsets of a part (64-bit) behind the names of the parts before it
__global__ void shard_rebase_names_kernel(uint64_t* out, const uint64_t* in, uint64_t n, uint64_t base) {
	const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = in[i] + base;
}
__global__ void shard_add_coverage_kernel(uint32_t* windows, uint8_t* starts, uint8_t* ends, const uint32_t* part_windows, const uint8_t* part_starts, const uint8_t* part_ends, uint64_t n) {
	const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { windows[i] += part_windows[i]; starts[i] |= part_starts[i]; ends[i] |= part_ends[i]; }
}
__global__ void shard_add_counts_kernel(unsigned long long* counts, const unsigned long long* part, uint32_t n) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) counts[i] += part[i];
}

// agpu_shard_merge when the names of the parts interleave: the order of std::string over the packed names ("QNAME,HI") of the merged batch
__global__ void packed_name_length_kernel(const uint64_t* name_offset, uint64_t n, uint32_t* longest) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i < n) atomicMax(longest, (uint32_t) (name_offset[i + 1] - name_offset[i]));
}
__global__ void packed_name_chunk_kernel(const char* names, const uint64_t* name_offset, const uint32_t* order, uint64_t n, uint32_t chunk, uint64_t* keys) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= n) return;
	const uint32_t row = order[i]; const uint64_t begin = name_offset[row]; const uint32_t length = (uint32_t) (name_offset[row + 1] - begin);
	uint64_t key = 0; // eight bytes from 8 * chunk on, big endian, zero padded: unsigned comparison of the chunks in order == std::string::compare
	for (uint32_t k = 0; k < 8; ++k) { const uint32_t at = 8 * chunk + k; key = key << 8 | (at < length ? (uint8_t) names[begin + at] : 0u); }
	keys[i] = key;
}
// rows i - 1 and i of the order: [0] counts equal names (a read name in two parts), new_group[i] says whether the QNAME (the name without ",HI...") changes
__global__ void packed_name_neighbours_kernel(const char* names, const uint64_t* name_offset, const uint32_t* order, uint64_t n, uint32_t* new_group, uint32_t* equal_names) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= n) return;
	if (i == 0) { new_group[0] = 0; return; }
	const uint32_t a = order[i - 1], b = order[i]; const uint64_t begin_a = name_offset[a], begin_b = name_offset[b]; const uint32_t length_a = (uint32_t) (name_offset[a + 1] - begin_a), length_b = (uint32_t) (name_offset[b + 1] - begin_b);
	bool same = length_a == length_b;
	for (uint32_t k = 0; same && k < length_a; ++k) same = names[begin_a + k] == names[begin_b + k];
	if (same) atomicAdd(equal_names, 1u);
	uint32_t qname_a = length_a, qname_b = length_b; // up to the last comma
	while (qname_a > 0 && names[begin_a + qname_a - 1] != ',') --qname_a;
	while (qname_b > 0 && names[begin_b + qname_b - 1] != ',') --qname_b;
	bool same_qname = qname_a == qname_b;
	for (uint32_t k = 0; same_qname && k < qname_a; ++k) same_qname = names[begin_a + k] == names[begin_b + k];
	new_group[i] = same_qname ? 0u : 1u;
}
// a part of a sample: where a new read name starts in the stream of the part, and the 128-bit keys of these names (ingest_core.hpp: qname_key128)
__global__ void qname_run_flag_kernel(IngestStream in, uint64_t n_records, uint8_t* flags) {
	const uint64_t r = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (r < n_records) flags[r] = starts_qname_run(in, (uint32_t) r) ? 1 : 0;
}
__global__ void qname_key_kernel(IngestStream in, const uint32_t* run_starts, uint64_t n_runs, uint64_t* keys) {
	const uint64_t k = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (k < n_runs) qname_key128(load_record(in, run_starts[k]), keys[2 * k], keys[2 * k + 1]);
}
__global__ void qname_low_kernel(const uint64_t* keys, uint64_t n, uint64_t* low) { const uint64_t k = blockIdx.x * (uint64_t) BLOCK + threadIdx.x; if (k < n) low[k] = keys[2 * k]; }
__global__ void qname_equal_kernel(const uint64_t* low_sorted, const uint32_t* index_sorted, const uint64_t* keys, uint64_t n, uint32_t* equal) {
	const uint64_t k = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (k > 0 && k < n && low_sorted[k] == low_sorted[k - 1] && keys[2 * (uint64_t) index_sorted[k] + 1] == keys[2 * (uint64_t) index_sorted[k - 1] + 1]) atomicAdd(equal, 1u);
}
__global__ void iota_kernel(uint32_t* out, uint64_t n) { const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x; if (i < n) out[i] = (uint32_t) i; }

// the windows of coverage_t from the prefix sums over the difference array of the replay (ingest_core.hpp: CoverageRange): window i of contig c is slot i + c; the counts in
// 32 bits (what a part of a sample hands on: agpu_shard_export) and clamped to the reference's 16 bits
__global__ void coverage_from_differences_kernel(const uint32_t* summed, const uint64_t* window_offset, uint32_t n_contigs, uint64_t n, uint32_t* counts, uint16_t* out) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= n) return;
	uint32_t low = 0, high = n_contigs; // the contig of window i: the last one with window_offset[c] <= i
	while (high - low > 1) { const uint32_t middle = (low + high) / 2; if (window_offset[middle] <= i) low = middle; else high = middle; }
	const uint32_t count = summed[i + low];
	counts[i] = count;
	out[i] = count > 65535u ? (uint16_t) 65535 : (uint16_t) count;
}
__global__ void coverage_clamp_kernel(const uint32_t* windows, uint64_t n, uint16_t* out) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i < n) out[i] = windows[i] > 65535u ? (uint16_t) 65535 : (uint16_t) windows[i];
}

// ---- rows of a resident batch for the host (names, CIGARs and sequences of the reads the output writer prints) -------------------------------------

__global__ void gather_sizes_kernel(BatchView b, const uint64_t* name_offset, const uint32_t* ids, uint64_t n, uint32_t* cigar_words, uint32_t* sequence_bytes, uint32_t* name_lengths) {
	const uint64_t k = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (k > n) return;
	if (k == n) { cigar_words[k] = 0; sequence_bytes[k] = 0; name_lengths[k] = 0; return; }
	row_sizes(b, name_offset, ids ? ids[k] : k, cigar_words[k], sequence_bytes[k], name_lengths[k]);
}

__global__ void gather_copy_kernel(BatchView b, const uint8_t* pristine_fbits, const uint8_t* const* pristine_abits, const uint64_t* name_offset, const char* names, const uint32_t* ids, uint64_t n,
                                   const uint64_t* cigar_base, const uint64_t* sequence_base, const uint64_t* name_base, PackTarget out) {
	const uint64_t k = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (k == n) { if (out.name_offset) out.name_offset[n] = name_base[n]; else out.row_name_offset[n] = (uint32_t) name_base[n]; }
	if (k >= n) return;
	copy_row(b, pristine_fbits, pristine_abits, name_offset, names, ids ? ids[k] : k, k, cigar_base[k], sequence_base[k], name_base[k], out);
}

// ---- detect_strandedness (source/read_stats.cpp:94-143): the first 100 informative split reads in name order ---------------------------------------

// 0 = not informative, 1 = informative, 3 = informative and matching the gene's strand
__global__ void strandedness_flag_kernel(BatchView b, AnnotationView ann, uint64_t first, uint64_t count, uint8_t* flags) {
	const uint64_t k = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (k >= count) return;
	flags[k] = strandedness_vote(b, ann, first + k);
}

// one thread: the first `wanted` informative fragments of the chunk, in order
__global__ void strandedness_take_kernel(const uint8_t* flags, uint64_t count, uint32_t wanted, uint32_t* counters) {
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	uint32_t seen = counters[IC_STRAND_COUNT], matching = counters[IC_STRAND_MATCHING];
	for (uint64_t k = 0; k < count && seen < wanted; ++k)
		if (flags[k]) { ++seen; if (flags[k] & 2) ++matching; }
	counters[IC_STRAND_COUNT] = seen; counters[IC_STRAND_MATCHING] = matching;
}

// ---- host side of the C ABI -----------------------------------------------------------------------------------------------------------------------

int grow_stream(agpu_ctx* ctx, uint64_t needed) {
	needed += 64; // (the readers of the stream -- record headers, the copies of bgzf_inflate_resolve_kernel -- load whole words that may end a few bytes behind the last byte)
	if (needed <= ctx->ingest_stream.capacity) { ctx->ingest_stream.bytes = needed; return AGPU_OK; }
	DeviceBuffer larger;
	const uint64_t doubled = ctx->ingest_stream.capacity * 2;
	if (!larger.allocate(std::max<uint64_t>(needed + (needed >> 3), std::max<uint64_t>(doubled, 64u << 20)))) { set_last_error("hipMalloc failed (BAM stream)"); return AGPU_ERR_NO_MEMORY; }
	if (ctx->piece_stream) { HIP_CHECK(hipStreamSynchronize(ctx->piece_stream)); HIP_CHECK(hipStreamSynchronize(ctx->piece_stream2)); } // (the pieces before this one are being unwrapped into the old one)
	if (ctx->ingest_stream_size > 0) HIP_CHECK(hipMemcpyAsync(larger.ptr, ctx->ingest_stream.ptr, ctx->ingest_stream_size, hipMemcpyDeviceToDevice, ctx->stream));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	if (ctx->ingest_progress.work) HIP_CHECK(hipStreamSynchronize(ctx->ingest_progress.work)); // (the windows in flight read the old one)
	ctx->ingest_stream.swap(larger);
	return AGPU_OK;
}

int wait_for_previous_push(agpu_ctx* ctx) {
	// push k was enqueued; the caller fills the buffer of push k - (host_buffers - 1) next: that piece must have left it when the caller gets control back
	const uint32_t behind = ctx->ingest_host_buffers - 1;
	if (ctx->ingest_pushes >= behind) HIP_CHECK(hipEventSynchronize(ctx->piece_copied[(ctx->ingest_pushes - behind) % AGPU_PIECE_SLOTS]));
	return AGPU_OK;
}

template <class Key> int sort_pairs(agpu_ctx* ctx, DeviceBuffer& scratch, Key* keys_in, Key* keys_out, uint32_t* values_in, uint32_t* values_out, uint64_t n, unsigned int end_bit, const char* label, bool counting_values) {
	size_t bytes = 0;
	hipStream_t s = ctx->stream;
	if (counting_values) HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, rocprim::counting_iterator<uint32_t>(0), values_out, n, 0, end_bit, s));
	else HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, values_in, values_out, n, 0, end_bit, s));
	if (bytes > scratch.capacity) ALLOC(scratch, bytes);
	KernelTimer timer(ctx, label, n * (uint64_t) (sizeof(Key) + 4) * 2);
	if (counting_values) HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in, keys_out, rocprim::counting_iterator<uint32_t>(0), values_out, n, 0, end_bit, s));
	else HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in, keys_out, values_in, values_out, n, 0, end_bit, s));
	return AGPU_OK;
}

// indices of the set flags, in order; *count on the device (one word of `counters`)
int select_flagged(agpu_ctx* ctx, DeviceBuffer& scratch, const uint8_t* flags, uint32_t* out, uint32_t* device_count, uint64_t n) {
	size_t bytes = 0;
	HIP_CHECK(rocprim::select(nullptr, bytes, rocprim::counting_iterator<uint32_t>(0), flags, out, device_count, n, ctx->stream));
	if (bytes > scratch.capacity) ALLOC(scratch, bytes);
	HIP_CHECK(rocprim::select(scratch.ptr, bytes, rocprim::counting_iterator<uint32_t>(0), flags, out, device_count, n, ctx->stream));
	return AGPU_OK;
}

int exclusive_sum_u64(agpu_ctx* ctx, DeviceBuffer& scratch, const uint32_t* in, uint64_t* out, uint64_t n) {
	size_t bytes = 0;
	HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, (uint64_t) 0, n, rocprim::plus<uint64_t>(), ctx->stream));
	if (bytes > scratch.capacity) ALLOC(scratch, bytes);
	HIP_CHECK(rocprim::exclusive_scan(scratch.ptr, bytes, in, out, (uint64_t) 0, n, rocprim::plus<uint64_t>(), ctx->stream));
	return AGPU_OK;
}

// ---- the front as the pieces arrive -----------------------------------------------------------------------------------------------------------------
// agpu_ingest_finish used to start when the last piece was in HBM: 0.9 s of kernels behind 1.5 s of copies at 10^8 fragments, the device idle while the file is fed.
// Now every pushed piece makes a window of the stream (IngestWindow, agpu_context.hpp) that goes through the front of the ingest -- record chain, record keys, runs of one
// name, the loop body of the reference per name -- on a stream of its own while the next pieces are copied; what is left for agpu_ingest_finish are the last windows and
// everything that needs all fragments (name order, pool offsets, the pack).  The runs of a name in the stream stand in for the groups that the sort by name key finds:
// the same groups in the same order as long as no name has records in two places, which the keys of all runs tell at the end (run_repeat_kernel); if one has -- or if a
// window meets anything else it cannot decide on its own -- the windows are abandoned and the whole stream is done the other way, as before.

const unsigned int INGEST_WINDOW_RING = 16, INGEST_WINDOW_WORDS = 8; // read-back words of a window: [0] segments that do not start where the one before ends, [1] records up to
                                                                     // the end of the window, [2] active records of the window, [3] runs that start in it, [4..5] where its last record ends
uint64_t g_window_bytes = 512u << 20; // a window is made when this much of the stream has arrived since the last one (10^8 fragments, step: 256 MB 4.1 s, 512 MB 3.89 s, 1 GB 3.97 s: profiles/r03n) ...
uint64_t g_window_margin = 1u << 20;  // ... and ends this far in front of the last byte that has: no record of it reaches behind (one that does: abandoned)

// a buffer that grows while windows behind it are in flight keeps its contents
int grow_keeping(agpu_ctx* ctx, DeviceBuffer& buffer, size_t needed, size_t used, size_t estimate) {
	if (needed == 0) needed = 16;
	if (buffer.ptr != nullptr && needed <= buffer.capacity) { if (needed > buffer.bytes) buffer.bytes = needed; return AGPU_OK; }
	DeviceBuffer larger;
	if (!larger.allocate(std::max(needed + needed / 2, estimate)) && !larger.allocate(needed)) { set_last_error("hipMalloc failed (tables of the ingest)"); return AGPU_ERR_NO_MEMORY; }
	hipStream_t s = ctx->ingest_progress.work;
	if (used > 0 && buffer.ptr != nullptr) HIP_CHECK(hipMemcpyAsync(larger.ptr, buffer.ptr, std::min(used, buffer.capacity), hipMemcpyDeviceToDevice, s));
	HIP_CHECK(hipStreamSynchronize(s));
	buffer.swap(larger);
	return AGPU_OK;
}

IngestStream window_stream(agpu_ctx* ctx, uint64_t size, uint64_t n_records) {
	IngestStream in;
	in.bytes = ctx->ingest_stream.as<uint8_t>(); in.size = size; in.record_offset = ctx->scratch("ingest.record_offset").as<uint64_t>(); in.n_records = n_records;
	in.n_targets = ctx->ingest_n_targets; in.tid_to_contig = ctx->ingest_tid_to_contig.as<uint32_t>();
	in.hit_index = ctx->scratch("ingest.hit_index").as<int32_t>(); // (null until the first window has parsed its records: nobody asks before)
	return in;
}

IngestContext replay_context(agpu_ctx* ctx, const IngestStream& in) {
	IngestContext context;
	context.stream = in; context.annotation = ctx->annotation; context.annotation.n_dummy = 0; context.genome = ctx->genome;
	context.coverage.n_contigs = ctx->genome.n_contigs; context.coverage.window_offset = ctx->coverage_window_offset.as<uint64_t>(); context.coverage.windows = ctx->coverage_windows32.as<uint32_t>();
	context.coverage.fragment_starts = ctx->coverage_fragment_starts.as<uint8_t>(); context.coverage.fragment_ends = ctx->coverage_fragment_ends.as<uint8_t>();
	if (getenv("ARRIBA_INGEST_SKIP_COVERAGE") != nullptr) context.coverage.windows = nullptr; // a measurement (the results are not the reference's): what the atomics on coverage_t cost the replay
	context.record_bits = ctx->scratch("ingest.record_bits").as<uint8_t>(); context.max_itd_length = ctx->ingest_max_itd_length; context.external_duplicate_marking = ctx->ingest_external_duplicate_marking;
	return context;
}

// enqueues the next step of a window; the words it leaves are read by window_note when the event behind them has passed
int window_step(agpu_ctx* ctx, IngestWindow& w) {
	IngestProgress& p = ctx->ingest_progress;
	hipStream_t s = p.work;
	uint32_t* words = ctx->scratch("ingest.window_words").as<uint32_t>() + (size_t) w.slot * INGEST_WINDOW_WORDS;
	uint32_t* device_counters = ctx->scratch("ingest.counters").as<uint32_t>();
	DeviceBuffer& rocprim_scratch = ctx->scratch("ingest.window_rocprim");
	const uint64_t base = ctx->ingest_first_record;
	const uint8_t* bytes = ctx->ingest_stream.as<uint8_t>();
	const uint64_t n = w.segment_end - w.segment_begin;
	const size_t estimated_records = (size_t) (p.size_hint / 128 + 4096);
	size_t temporary = 0;
	switch (w.enqueued + 1) {
	case 1: { // the record chain of the window's segments, behind the chain of the windows before
		HIP_CHECK(hipMemsetAsync(words, 0, INGEST_WINDOW_WORDS * 4, s));
		if (n == 0) break; // (a stream without a record: the last window of an empty file)
		DeviceBuffer& first = ctx->scratch("ingest.segment_first"); DeviceBuffer& end = ctx->scratch("ingest.segment_end"); DeviceBuffer& end_before = ctx->scratch("ingest.segment_end_before");
		DeviceBuffer& count = ctx->scratch("ingest.segment_count"); DeviceBuffer& record_base = ctx->scratch("ingest.segment_base"); DeviceBuffer& mismatch = ctx->scratch("ingest.segment_mismatch");
		const size_t segments = (size_t) w.segment_end + 1, done = (size_t) w.segment_begin + 1, estimate = (size_t) (p.size_hint / SEGMENT_BYTES + 16);
		TRY(grow_keeping(ctx, first, segments * 8, done * 8, estimate * 8)); TRY(grow_keeping(ctx, end, segments * 8, done * 8, estimate * 8)); TRY(grow_keeping(ctx, end_before, segments * 8, 0, estimate * 8));
		TRY(grow_keeping(ctx, count, segments * 4, done * 4, estimate * 4)); TRY(grow_keeping(ctx, record_base, segments * 4, done * 4, estimate * 4)); TRY(grow_keeping(ctx, mismatch, segments, done, estimate));
		{ KernelTimer timer(ctx, "segment_guess_kernel", n * SEGMENT_BYTES, s);
		  segment_guess_kernel<<<grid_for(n), BLOCK, 0, s>>>(bytes, w.avail, base, w.segment_begin, w.segment_end, ctx->ingest_n_targets, first.as<uint64_t>(), end.as<uint64_t>(), count.as<uint32_t>()); }
		// (the loop of the other way reads the number of mismatches after every pass; here three passes are made whether they are needed or not -- a pass over segments
		// that fit is a few microseconds -- and a window that still has a mismatch then gives up)
		for (int pass = 0; pass < 3; ++pass) {
			segment_check_kernel<<<tally_grid(n, BLOCK), BLOCK, 0, s>>>(first.as<uint64_t>(), end.as<uint64_t>(), w.segment_begin, w.segment_end, mismatch.as<uint8_t>(), words + 6);
			const uint64_t from = w.segment_begin > 0 ? w.segment_begin - 1 : 0;
			HIP_CHECK(hipMemcpyAsync(end_before.as<uint64_t>() + from, end.as<uint64_t>() + from, (w.segment_end - from) * 8, hipMemcpyDeviceToDevice, s));
			segment_repair_kernel<<<grid_for(n), BLOCK, 0, s>>>(bytes, w.avail, base, w.segment_begin, w.segment_end, mismatch.as<uint8_t>(), end_before.as<uint64_t>(), first.as<uint64_t>(), end.as<uint64_t>(), count.as<uint32_t>());
		}
		segment_check_kernel<<<tally_grid(n, BLOCK), BLOCK, 0, s>>>(first.as<uint64_t>(), end.as<uint64_t>(), w.segment_begin, w.segment_end, mismatch.as<uint8_t>(), words + 0);
		HIP_CHECK(hipMemsetAsync(count.as<uint32_t>() + w.segment_end, 0, 4, s));
		HIP_CHECK(rocprim::exclusive_scan(nullptr, temporary, count.as<uint32_t>() + w.segment_begin, record_base.as<uint32_t>() + w.segment_begin, 0u, n + 1, rocprim::plus<uint32_t>(), s));
		if (temporary > rocprim_scratch.capacity) ALLOC(rocprim_scratch, temporary);
		HIP_CHECK(rocprim::exclusive_scan(rocprim_scratch.ptr, temporary, count.as<uint32_t>() + w.segment_begin, record_base.as<uint32_t>() + w.segment_begin, 0u, n + 1, rocprim::plus<uint32_t>(), s));
		uint32_t* records_before = ctx->scratch("ingest.window_state").as<uint32_t>();
		window_carry_kernel<<<grid_for(n + 1), BLOCK, 0, s>>>(record_base.as<uint32_t>(), w.segment_begin, w.segment_end, records_before);
		window_summary_kernel<<<1, 1, 0, s>>>(record_base.as<uint32_t>(), end.as<uint64_t>(), w.segment_end, records_before, words);
		break; }
	case 2: { // where the records start; status and name key of every record; the active ones, in the order of the stream
		const uint64_t records = w.record_end - w.record_begin;
		w.active_begin = p.active; // (the same step of the window before has been read: windows_advance)
		if (records == 0) break;
		DeviceBuffer& record_offset = ctx->scratch("ingest.record_offset"); DeviceBuffer& keys = ctx->scratch("ingest.keys"); DeviceBuffer& record_bits = ctx->scratch("ingest.record_bits"); DeviceBuffer& active_records = ctx->scratch("ingest.sorted_records");
		TRY(grow_keeping(ctx, record_offset, w.record_end * 8, w.record_begin * 8, estimated_records * 8)); TRY(grow_keeping(ctx, keys, w.record_end * 8, w.record_begin * 8, estimated_records * 8));
		TRY(grow_keeping(ctx, ctx->scratch("ingest.hit_index"), w.record_end * 4, w.record_begin * 4, estimated_records * 4));
		TRY(grow_keeping(ctx, record_bits, w.record_end, w.record_begin, estimated_records)); TRY(grow_keeping(ctx, active_records, (w.active_begin + records) * 4, w.active_begin * 4, estimated_records * 4));
		{ KernelTimer timer(ctx, "segment_emit_kernel", records * 8, s);
		  segment_emit_kernel<<<grid_for(n), BLOCK, 0, s>>>(bytes, w.avail, base, w.segment_begin, w.segment_end, ctx->scratch("ingest.segment_first").as<uint64_t>(), ctx->scratch("ingest.segment_base").as<uint32_t>(), record_offset.as<uint64_t>()); }
		const IngestStream in = window_stream(ctx, w.avail, w.record_end);
		{ KernelTimer timer(ctx, "record_parse_kernel", records * (8 + 64 + 9), s);
		  record_parse_kernel<<<tally_grid(records, BLOCK) * 8, BLOCK, 0, s>>>(in, ctx->genome, 0, w.record_begin, keys.as<uint64_t>(), record_bits.as<uint8_t>(), ctx->scratch("ingest.hit_index").as<int32_t>(), device_counters); }
		auto flags = rocprim::make_transform_iterator(record_bits.as<uint8_t>() + w.record_begin, RecordIsActive());
		HIP_CHECK(rocprim::select(nullptr, temporary, rocprim::counting_iterator<uint32_t>((uint32_t) w.record_begin), flags, active_records.as<uint32_t>() + w.active_begin, words + 2, records, s));
		if (temporary > rocprim_scratch.capacity) ALLOC(rocprim_scratch, temporary);
		HIP_CHECK(rocprim::select(rocprim_scratch.ptr, temporary, rocprim::counting_iterator<uint32_t>((uint32_t) w.record_begin), flags, active_records.as<uint32_t>() + w.active_begin, words + 2, records, s));
		break; }
	case 3: { // where the runs of one name start
		const uint64_t active = w.active_end - w.active_begin;
		w.head_begin = p.heads;
		if (active == 0) break;
		DeviceBuffer& head = ctx->scratch("ingest.head"); DeviceBuffer& run_begin = ctx->scratch("ingest.group_begin");
		TRY(grow_keeping(ctx, head, w.active_end, 0, estimated_records)); TRY(grow_keeping(ctx, run_begin, (w.head_begin + active + 1) * 4, w.head_begin * 4, estimated_records * 2));
		{ KernelTimer timer(ctx, "run_head_kernel", active * (8 + 4 + 1), s);
		  run_head_kernel<<<grid_for(active), BLOCK, 0, s>>>(ctx->scratch("ingest.keys").as<uint64_t>(), ctx->scratch("ingest.sorted_records").as<uint32_t>(), w.active_begin, w.active_end, head.as<uint8_t>()); }
		HIP_CHECK(rocprim::select(nullptr, temporary, rocprim::counting_iterator<uint32_t>((uint32_t) w.active_begin), head.as<uint8_t>() + w.active_begin, run_begin.as<uint32_t>() + w.head_begin, words + 3, active, s));
		if (temporary > rocprim_scratch.capacity) ALLOC(rocprim_scratch, temporary);
		HIP_CHECK(rocprim::select(rocprim_scratch.ptr, temporary, rocprim::counting_iterator<uint32_t>((uint32_t) w.active_begin), head.as<uint8_t>() + w.active_begin, run_begin.as<uint32_t>() + w.head_begin, words + 3, active, s));
		break; }
	case 4: { // the runs that are complete (all but the one the window ends in, which the next window may continue): equal keys are equal names; the loop body of the reference
		const uint64_t first_group = p.groups_done, n_groups = w.last ? w.head_end : (w.head_end > 0 ? w.head_end - 1 : 0);
		if (n_groups <= first_group) break;
		DeviceBuffer& group_first = ctx->scratch("ingest.group_first"); DeviceBuffer& group_count = ctx->scratch("ingest.group_count");
		DeviceBuffer& plain = ctx->scratch("ingest.plain_plans"); DeviceBuffer& itd = ctx->scratch("ingest.itd_plans"); DeviceBuffer& valid = ctx->scratch("ingest.valid"); DeviceBuffer& sizes = ctx->scratch("ingest.sizes");
		const size_t estimated_groups = estimated_records / 3;
		TRY(grow_keeping(ctx, group_first, n_groups * 4, first_group * 4, estimated_groups * 4)); TRY(grow_keeping(ctx, group_count, n_groups * 4, first_group * 4, estimated_groups * 4));
		TRY(grow_keeping(ctx, plain, n_groups * sizeof(FragmentPlan), first_group * sizeof(FragmentPlan), estimated_groups * sizeof(FragmentPlan))); TRY(grow_keeping(ctx, itd, n_groups * sizeof(TandemPlan), first_group * sizeof(TandemPlan), estimated_groups * sizeof(TandemPlan)));
		TRY(grow_keeping(ctx, valid, 2 * n_groups, 2 * first_group, 2 * estimated_groups)); TRY(grow_keeping(ctx, sizes, 2 * n_groups * sizeof(FragmentSizes), 2 * first_group * sizeof(FragmentSizes), 2 * estimated_groups * sizeof(FragmentSizes)));
		const uint32_t* active_records = ctx->scratch("ingest.sorted_records").as<uint32_t>(); const uint32_t* run_begin = ctx->scratch("ingest.group_begin").as<uint32_t>();
		const uint64_t groups = n_groups - first_group;
		run_close_kernel<<<grid_for(groups), BLOCK, 0, s>>>(active_records, run_begin, (uint32_t) first_group, (uint32_t) n_groups, (uint32_t) w.head_end, w.active_end, group_first.as<uint32_t>(), group_count.as<uint32_t>());
		const IngestStream in = window_stream(ctx, w.avail, w.record_end);
		DeviceBuffer& qname_differs = ctx->scratch("ingest.qname_differs");
		TRY(grow_keeping(ctx, qname_differs, n_groups * 4, first_group * 4, estimated_groups * 4));
		{ KernelTimer timer(ctx, "group_replay_kernel", n * SEGMENT_BYTES, s);
		  group_replay_kernel<true><<<grid_for(groups), BLOCK, 0, s>>>(replay_context(ctx, in), active_records, run_begin, group_count.as<uint32_t>(), (uint32_t) first_group, (uint32_t) n_groups, plain.as<FragmentPlan>(), itd.as<TandemPlan>(), valid.as<uint8_t>(),
		                                                              sizes.as<FragmentSizes>(), ctx->ingest_viral_counts.as<unsigned long long>(), qname_differs.as<uint32_t>(), device_counters); }
		{ KernelTimer timer(ctx, "run_itd_order_kernel", groups * 2, s);
		  run_itd_order_kernel<<<grid_for(groups), BLOCK, 0, s>>>(in, group_first.as<uint32_t>(), valid.as<uint8_t>(), (uint32_t) first_group, (uint32_t) n_groups, device_counters); }
		p.groups_done = n_groups; p.touched = true;
		break; }
	}
	HIP_CHECK(hipMemcpyAsync(p.host_words + (size_t) w.slot * INGEST_WINDOW_WORDS, words, INGEST_WINDOW_WORDS * 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipEventRecord(w.readback, s));
	++w.enqueued;
	return AGPU_OK;
}

// the words of the step enqueued last have arrived
void window_note(agpu_ctx* ctx, IngestWindow& w) {
	IngestProgress& p = ctx->ingest_progress;
	const uint32_t* words = p.host_words + (size_t) w.slot * INGEST_WINDOW_WORDS;
	w.known = w.enqueued;
	switch (w.known) {
	case 1: {
		w.record_begin = p.records;
		if (w.segment_end == w.segment_begin) { w.record_end = p.records; break; }
		const uint64_t last_end = (uint64_t) words[4] | (uint64_t) words[5] << 32;
		if (words[0] != 0 || last_end >= OFFSET_BROKEN || (w.last && last_end != w.avail)) { p.abandoned = true; break; } // (a damaged file, too: the other way says so)
		w.record_end = words[1]; p.records = w.record_end;
		break; }
	case 2: w.active_end = w.active_begin + (w.record_end > w.record_begin ? words[2] : 0); p.active = w.active_end; break;
	case 3: w.head_end = w.head_begin + (w.active_end > w.active_begin ? words[3] : 0); p.heads = w.head_end; break;
	}
}

// moves the windows on as far as the words that have arrived allow; wait: until every window is through (agpu_ingest_finish), otherwise nobody waits for the device
int windows_advance(agpu_ctx* ctx, bool wait) {
	IngestProgress& p = ctx->ingest_progress;
	while (!p.abandoned && !p.windows.empty()) {
		bool moved = false;
		for (size_t k = 0; k < p.windows.size() && !p.abandoned; ++k) {
			IngestWindow& w = p.windows[k];
			if (w.enqueued > w.known) {
				// the words of a step are taken in window order: step 1 of window k takes its first record from what window k-1 has noted (both can be in flight at once, and two
				// hipEventQuery calls apart a thread that was descheduled in between may see k ready and k-1 not yet)
				if (k > 0 && p.windows[k - 1].known < w.enqueued) break;
				const hipError_t state = hipEventQuery(w.readback);
				if (state == hipErrorNotReady) break; // (the windows behind it queue behind it on the same stream)
				if (state != hipSuccess) { set_last_error(std::string("hipEventQuery: ") + hipGetErrorString(state)); return AGPU_ERR_DEVICE; }
				window_note(ctx, w);
				moved = true;
				if (p.abandoned) break;
			}
			// a step needs what the same step of the window before has left (records, active records, runs so far)
			if (w.known == w.enqueued && w.enqueued < 4 && (k == 0 || p.windows[k - 1].known > w.enqueued || (w.enqueued == 3 && p.windows[k - 1].enqueued == 4))) { TRY(window_step(ctx, w)); moved = true; }
		}
		while (!p.windows.empty() && p.windows.front().known == 4) { p.events.push_back(p.windows.front().readback); p.windows.pop_front(); moved = true; }
		if (moved) continue;
		if (!wait) break;
		for (size_t k = 0; k < p.windows.size(); ++k) if (p.windows[k].enqueued > p.windows[k].known) { HIP_CHECK(hipEventSynchronize(p.windows[k].readback)); break; }
	}
	return AGPU_OK;
}

// a new window over what has arrived (last: over everything, made by agpu_ingest_finish)
int window_make(agpu_ctx* ctx, bool last) {
	IngestProgress& p = ctx->ingest_progress;
	const uint64_t size = ctx->ingest_stream_size, base = ctx->ingest_first_record;
	uint64_t segment_end = 0;
	if (last) segment_end = size > base ? (size - base + SEGMENT_BYTES - 1) / SEGMENT_BYTES : 0;
	else if (size > base + g_window_margin) segment_end = (size - base - g_window_margin) / SEGMENT_BYTES;
	if (!last && segment_end <= p.segments_done) return AGPU_OK;
	while (p.windows.size() + 2 > INGEST_WINDOW_RING) { // (the device is further behind than the ring is long: wait for the oldest window)
		IngestWindow& oldest = p.windows.front();
		if (oldest.enqueued > oldest.known) HIP_CHECK(hipEventSynchronize(oldest.readback));
		TRY(windows_advance(ctx, false));
		if (p.abandoned) return AGPU_OK;
	}
	IngestWindow w;
	w.avail = size; w.segment_begin = p.segments_done; w.segment_end = std::max(segment_end, p.segments_done); w.last = last; w.slot = p.next_slot; p.next_slot = (p.next_slot + 1) % INGEST_WINDOW_RING;
	if (!p.events.empty()) { w.readback = p.events.back(); p.events.pop_back(); } else HIP_CHECK(hipEventCreateWithFlags(&w.readback, hipEventDisableTiming));
	p.segments_done = w.segment_end; p.window_bytes = size; ++p.windows_made;
	if (ctx->ingest_pushes > 0) HIP_CHECK(hipStreamWaitEvent(p.work, ctx->piece_ready[(ctx->ingest_pushes - 1) % AGPU_PIECE_SLOTS], 0)); // the bytes of the window are there
	p.windows.push_back(w);
	if (p.windows.size() == 1 || p.windows[p.windows.size() - 2].enqueued >= 1) TRY(window_step(ctx, p.windows.back())); // (the chain needs nothing from the host)
	return AGPU_OK;
}

// behind a push: a window over the new bytes if enough have arrived, and whatever the windows in flight can do next
int windows_after_push(agpu_ctx* ctx) {
	IngestProgress& p = ctx->ingest_progress;
	if (!p.on || p.abandoned) return AGPU_OK;
	TRY(windows_advance(ctx, false));
	if (!p.abandoned && ctx->ingest_stream_size - p.window_bytes >= g_window_bytes) TRY(window_make(ctx, false));
	return AGPU_OK;
}

void fill_pack_target(agpu_ctx* ctx, PackTarget& out) {
	out.n_aln = ctx->n_aln.as<uint8_t>(); out.fbits = ctx->fbits.as<uint8_t>(); out.group = ctx->group.as<uint32_t>();
	for (int k = 0; k < 3; ++k) {
		out.contig[k] = ctx->contig[k].as<uint16_t>(); out.start[k] = ctx->start[k].as<int32_t>(); out.end[k] = ctx->end[k].as<int32_t>(); out.abits[k] = ctx->abits[k].as<uint8_t>();
		out.cigar_offset[k] = ctx->cigar_offset[k].as<uint32_t>(); out.cigar_count[k] = ctx->cigar_count[k].as<uint16_t>();
	}
	out.cigar_pool = ctx->cigar_pool.as<uint32_t>();
	for (int k = 0; k < 2; ++k) { out.seq_offset[k] = ctx->seq_offset[k].as<uint32_t>(); out.seq_length[k] = ctx->seq_length[k].as<uint32_t>(); }
	out.seq_pool = ctx->seq_pool.as<uint8_t>(); out.name_offset = ctx->name_offset.as<uint64_t>(); out.row_name_offset = nullptr; out.names = ctx->names.as<char>();
}

}

bool agpu::release_ingest_buffers(agpu_ctx* ctx) {
	bool released = ctx->ingest_stream.ptr != nullptr;
	ctx->ingest_stream.release(); for (int k = 0; k < AGPU_PIECE_SLOTS; ++k) ctx->ingest_raw[k].release();
	if (!ctx->ingest_part_of_sample) ctx->coverage_windows32.release(); // (a part of a sample hands the windows on as they are: agpu_shard_export)
	static const char* const temporary[] = { "ingest.record_offset", "ingest.keys", "ingest.keys_sorted", "ingest.record_bits", "ingest.sorted_records", "ingest.head", "ingest.group_start", "ingest.first_flags", "ingest.stream_rank", "ingest.group_first", "ingest.group_begin", "ingest.group_count", "ingest.plain_plans", "ingest.itd_plans",
		"ingest.valid", "ingest.sizes", "ingest.refs", "ingest.order", "ingest.order_keys", "ingest.order_keys_sorted", "ingest.cigar_words", "ingest.sequence_bytes", "ingest.name_lengths", "ingest.new_group", "ingest.cigar_base",
		"ingest.sequence_base", "ingest.name_base", "ingest.group_id", "ingest.qname_differs", "ingest.qname_run", "ingest.run_keys", "ingest.run_keys_sorted", "ingest.window_rocprim", "ingest.segment_first", "ingest.segment_end", "ingest.segment_end_before", "ingest.segment_count", "ingest.segment_base", "ingest.segment_mismatch", "ingest.rocprim", "ingest.coverage_summed", "ingest.hit_index" };
	for (size_t k = 0; k < sizeof(temporary) / sizeof(temporary[0]); ++k) { DeviceBuffer& buffer = ctx->scratch(temporary[k]); if (buffer.ptr != nullptr) released = true; buffer.release(); }
	return released;
}

extern "C" {

void* agpu_host_alloc(size_t bytes) {
	void* pointer = nullptr;
	if (hipHostMalloc(&pointer, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) { set_last_error("hipHostMalloc failed"); return nullptr; }
	return pointer;
}
void agpu_host_free(void* pointer) { if (pointer) (void) hipHostFree(pointer); }

static int gather_rows_prepare(agpu_ctx* ctx, const uint32_t* fragments, uint64_t n, uint64_t* pool_sizes);

int agpu_ingest_begin(agpu_ctx* ctx, const agpu_ingest_config* config) {
	if (!ctx || !config) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	if (!ctx->have_annotation || !ctx->have_genome) { set_last_error("annotation and genome must be uploaded before the ingest"); return AGPU_ERR_INVALID; }
	if (config->n_contigs != ctx->genome.n_contigs) { set_last_error("the genome view on the device does not hold the contigs of the BAM header (upload it after the header was parsed)"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	{ std::lock_guard<std::mutex> lock(ctx->profile_mutex); ctx->failed_launch.clear(); } // (a new sample)
	if (!ctx->piece_stream) {
		int least = 0, greatest = 0;
		HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
		const char* priorities = getenv("ARRIBA_STREAM_PRIORITIES");
		const bool pieces_first = priorities == nullptr || strcmp(priorities, "stages") != 0;
		HIP_CHECK(hipStreamCreateWithPriority(&ctx->piece_stream, hipStreamNonBlocking, pieces_first ? greatest : (least + greatest) / 2)); // (in front of the kernels of the windows: the feed waits for these; behind the stages of the other lane of a session)
		HIP_CHECK(hipStreamCreateWithPriority(&ctx->piece_stream2, hipStreamNonBlocking, pieces_first ? greatest : (least + greatest) / 2)); // (deflated pieces take the two in turn: see agpu_ingest_push_bgzf)
		for (int k = 0; k < AGPU_PIECE_SLOTS; ++k) { HIP_CHECK(hipEventCreateWithFlags(&ctx->piece_copied[k], hipEventDisableTiming | hipEventBlockingSync)); /* (the thread that waits for a copy leaves its core to the readers of the file) */ HIP_CHECK(hipEventCreateWithFlags(&ctx->piece_ready[k], hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&ctx->piece_done[k], hipEventDisableTiming)); }
	}
	if (config->host_buffers > AGPU_PIECE_SLOTS) { set_last_error("agpu_ingest_config.host_buffers: at most 4"); return AGPU_ERR_INVALID; }
	ctx->ingest_host_buffers = config->host_buffers < 2 ? 2 : config->host_buffers;
	ctx->ingest_n_targets = config->n_targets; ctx->ingest_first_record = config->first_record_offset; ctx->ingest_stream_size = 0; ctx->ingest_pushes = 0; ctx->ingest_deflated_pieces = false;
	{ const char* knob = getenv("ARRIBA_VERIFY_CRC"); ctx->ingest_verify_crc = !(knob != nullptr && knob[0] == '0'); } // (the stored blocks are checked as htslib checks them; "0": a measurement without)
	ALLOC(ctx->scratch("ingest.crc_mismatches"), 8); // [0] blocks whose payload does not give the CRC-32 of their trailer, [1] deflated blocks that did not decode (read whatever ARRIBA_VERIFY_CRC says)
	if (ctx->ingest_verify_crc && ctx->scratch("ingest.crc_tables").ptr == nullptr) {
		ALLOC(ctx->scratch("ingest.crc_tables"), sizeof(Crc32Tables));
		static Crc32Tables tables; static bool made = false;
		if (!made) { crc32_make_tables(tables); made = true; }
		HIP_CHECK(hipMemcpy(ctx->scratch("ingest.crc_tables").ptr, &tables, sizeof(tables), hipMemcpyHostToDevice));
	}
	HIP_CHECK(hipMemsetAsync(ctx->scratch("ingest.crc_mismatches").ptr, 0, 8, s));
	ctx->ingest_external_duplicate_marking = config->external_duplicate_marking; ctx->ingest_max_itd_length = config->max_itd_length; ctx->ingest_part_of_sample = config->part_of_sample != 0;
	ALLOC(ctx->ingest_tid_to_contig, std::max<size_t>(config->n_targets, 1) * 4);
	if (config->n_targets) HIP_CHECK(hipMemcpyAsync(ctx->ingest_tid_to_contig.ptr, config->tid_to_contig, (size_t) config->n_targets * 4, hipMemcpyHostToDevice, s));
	ctx->host_coverage_window_offset.assign(config->coverage_window_offset, config->coverage_window_offset + config->n_contigs + 1);
	const uint64_t windows = ctx->host_coverage_window_offset.back();
	ALLOC(ctx->coverage_window_offset, ((size_t) config->n_contigs + 1) * 8); ALLOC(ctx->coverage_windows32, std::max<uint64_t>(coverage_difference_slots(windows, config->n_contigs), 1) * 4); ALLOC(ctx->coverage_windows, std::max<uint64_t>(windows, 1) * 2);
	ALLOC(ctx->coverage_fragment_starts, std::max<uint64_t>(windows, 1)); ALLOC(ctx->coverage_fragment_ends, std::max<uint64_t>(windows, 1));
	HIP_CHECK(hipMemcpyAsync(ctx->coverage_window_offset.ptr, config->coverage_window_offset, ((size_t) config->n_contigs + 1) * 8, hipMemcpyHostToDevice, s));
	HIP_CHECK(hipMemsetAsync(ctx->coverage_windows32.ptr, 0, std::max<uint64_t>(coverage_difference_slots(windows, config->n_contigs), 1) * 4, s));
	HIP_CHECK(hipMemsetAsync(ctx->coverage_fragment_starts.ptr, 0, std::max<uint64_t>(windows, 1), s));
	HIP_CHECK(hipMemsetAsync(ctx->coverage_fragment_ends.ptr, 0, std::max<uint64_t>(windows, 1), s));
	ALLOC(ctx->ingest_viral_counts, std::max<size_t>(config->n_contigs, 1) * 8);
	HIP_CHECK(hipMemsetAsync(ctx->ingest_viral_counts.ptr, 0, std::max<size_t>(config->n_contigs, 1) * 8, s));
	if (config->stream_size_hint > 0) TRY(grow_stream(ctx, config->stream_size_hint));
	{ // the front of the ingest while the pieces arrive (see "the front as the pieces arrive"); a part of a sample is done at the end, the way it was
		IngestProgress& p = ctx->ingest_progress;
		for (size_t k = 0; k < p.windows.size(); ++k) p.events.push_back(p.windows[k].readback); // (an ingest that was begun and never finished)
		p.windows.clear();
		const char* knob = getenv("ARRIBA_INGEST_WINDOWS"); // "0": everything behind the last piece (the way of round 2; for measurements and for the tests of that way); "bytes,margin": smaller windows (tests)
		p.on = !ctx->ingest_part_of_sample && !(knob != nullptr && knob[0] == '0' && knob[1] == 0);
		g_window_bytes = 512u << 20; g_window_margin = 1u << 20;
		if (knob != nullptr && strchr(knob, ',') != nullptr) { g_window_bytes = strtoull(knob, nullptr, 10); g_window_margin = std::max<uint64_t>(strtoull(strchr(knob, ',') + 1, nullptr, 10), 1); }
		p.abandoned = false; p.touched = false; p.next_slot = 0; p.window_bytes = 0; p.segments_done = 0; p.records = 0; p.active = 0; p.heads = 0; p.groups_done = 0; p.windows_made = 0; p.size_hint = config->stream_size_hint;
		if (p.on) {
			if (!p.work) {
				int least = 0, greatest = 0;
				HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
				HIP_CHECK(hipStreamCreateWithPriority(&p.work, hipStreamNonBlocking, least)); // (behind the copies and the unwrap kernels of the pieces, which the feed waits for)
				HIP_CHECK(hipHostMalloc((void**) &p.host_words, INGEST_WINDOW_RING * INGEST_WINDOW_WORDS * 4, hipHostMallocDefault));
			}
			ALLOC(ctx->scratch("ingest.window_words"), INGEST_WINDOW_RING * INGEST_WINDOW_WORDS * 4); ALLOC(ctx->scratch("ingest.window_state"), 16); ALLOC(ctx->scratch("ingest.counters"), IC_COUNT * 4);
			HIP_CHECK(hipMemsetAsync(ctx->scratch("ingest.window_state").ptr, 0, 16, s));
			HIP_CHECK(hipMemsetAsync(ctx->scratch("ingest.counters").ptr, 0, IC_COUNT * 4, s));
		}
	}
	HIP_CHECK(hipStreamSynchronize(s));
	ctx->ingest_active = true; ctx->have_batch = false; ctx->have_coverage = false;
	return AGPU_OK;
}

int agpu_ingest_push(agpu_ctx* ctx, const void* bytes, size_t size) {
	if (!ctx || !ctx->ingest_active) { set_last_error("agpu_ingest_begin must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	if (size > 0) {
		TRY(grow_stream(ctx, ctx->ingest_stream_size + size));
		HIP_CHECK(hipMemcpyAsync(ctx->ingest_stream.as<uint8_t>() + ctx->ingest_stream_size, bytes, size, hipMemcpyHostToDevice, ctx->stream));
		ctx->ingest_stream_size += size;
	}
	{ const unsigned int slot = ctx->ingest_pushes % AGPU_PIECE_SLOTS;
	  HIP_CHECK(hipEventRecord(ctx->piece_copied[slot], ctx->stream)); HIP_CHECK(hipEventRecord(ctx->piece_ready[slot], ctx->stream)); }
	TRY(wait_for_previous_push(ctx));
	++ctx->ingest_pushes;
	return windows_after_push(ctx);
}

int agpu_ingest_push_bgzf(agpu_ctx* ctx, const void* raw, size_t raw_size, const agpu_bgzf_block* blocks, uint32_t n_blocks, size_t stream_bytes) {
	if (!ctx || !ctx->ingest_active) { set_last_error("agpu_ingest_begin must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const unsigned int slot = ctx->ingest_pushes % AGPU_PIECE_SLOTS;
	hipStream_t pieces = ctx->piece_stream;
	if (n_blocks > 0 && raw_size > 0) {
		TRY(grow_stream(ctx, ctx->ingest_stream_size + stream_bytes));
		const bool deflated = blocks[0].isize != 0; // (all blocks of a piece are of one kind: include/arriba_gpu.h)
		// Deflated pieces take two streams in turn.  Pass 1 of the decoder holds 48 blocks per CU (its tables fill the LDS) and a piece is ~17 000 blocks: the second round of a piece
		// fills a third of the device, and passes 2 and the CRC leave most of its LDS-bound slots idle anyway -- with the next piece on the other stream its pass 1 runs in those gaps
		// (profiles/r05b: 20.7 ms per piece for pass 1 alone on one stream, 10.6 ms for a round).  The windows of the ingest wait for the LAST piece's event only: a piece is
		// reported ready behind the one in front of it.  Parts of a file (one spill buffer) and ARRIBA_INFLATE_STREAMS=1 stay on one stream.
		static const bool two_streams = !(getenv("ARRIBA_INFLATE_STREAMS") != nullptr && atoi(getenv("ARRIBA_INFLATE_STREAMS")) == 1);
		const int set = deflated && two_streams && !ctx->ingest_part_of_sample ? (int) (ctx->ingest_pushes & 1u) : 0;
		if (set == 1) pieces = ctx->piece_stream2;
		ALLOC(ctx->ingest_raw[slot], raw_size + 256); ALLOC(ctx->ingest_blocks[slot], (size_t) n_blocks * sizeof(agpu_bgzf_block)); // (+ 256: the bit readers of the inflate kernels load ahead of what they decode: inflate_fast_core.hpp)
		if (deflated) ALLOC(ctx->scratch("ingest.inflate_spill"), 2 * 65536);
		HIP_CHECK(hipStreamWaitEvent(s, ctx->piece_done[slot], 0)); // (the piece that lay in this buffer is unwrapped and checked; an event that was never recorded does not hold anybody up)
		HIP_CHECK(hipMemcpyAsync(ctx->ingest_raw[slot].ptr, raw, raw_size, hipMemcpyHostToDevice, s));
		HIP_CHECK(hipMemcpyAsync(ctx->ingest_blocks[slot].ptr, blocks, (size_t) n_blocks * sizeof(agpu_bgzf_block), hipMemcpyHostToDevice, s));
		HIP_CHECK(hipEventRecord(ctx->piece_copied[slot], s));
		HIP_CHECK(hipStreamWaitEvent(pieces, ctx->piece_copied[slot], 0));
		if (deflated) {
			ctx->ingest_deflated_pieces = true;
			uint8_t* target = ctx->ingest_stream.as<uint8_t>() + ctx->ingest_stream_size;
			unsigned int* failures = ctx->scratch("ingest.crc_mismatches").as<unsigned int>() + 1;
			static const char* way = getenv("ARRIBA_INFLATE"); // "wave": the one-wavefront-per-block decoder of round 4 for every block; "16" / "24": other numbers of blocks per wavefront in pass 1 (measurements)
			if (way != nullptr && strcmp(way, "wave") == 0) {
				KernelTimer timer(ctx, "bgzf_inflate_kernel", (uint64_t) raw_size + stream_bytes, pieces);
				bgzf_inflate_kernel<<<n_blocks, 64, 0, pieces>>>(ctx->ingest_raw[slot].as<uint8_t>(), ctx->ingest_blocks[slot].as<agpu_bgzf_block>(), target, ctx->scratch("ingest.inflate_spill").as<uint8_t>(), nullptr, failures);
			} else {
				// (the kernels of the pieces of one stream run one after the other: one set of notes per stream serves all slots)
				DeviceBuffer& notes = ctx->scratch(set ? "ingest.inflate_notes2" : "ingest.inflate_notes"); DeviceBuffer& note_count = ctx->scratch(set ? "ingest.inflate_note_count2" : "ingest.inflate_note_count"); DeviceBuffer& status = ctx->scratch(set ? "ingest.inflate_status2" : "ingest.inflate_status");
				if ((size_t) n_blocks * INFLATE_MATCH_CAPACITY * 8 > notes.capacity || (size_t) n_blocks * 4 > status.capacity) {
					HIP_CHECK(hipStreamSynchronize(pieces)); // (the pieces before this one read the buffers that are about to be replaced)
					ALLOC(notes, (size_t) n_blocks * INFLATE_MATCH_CAPACITY * 8); ALLOC(note_count, (size_t) n_blocks * 4); ALLOC(status, (size_t) n_blocks * 4);
				}
				const int lanes = way != nullptr && atoi(way) > 0 ? atoi(way) : 20; // (20 lanes x 2 032 bytes of tables: four such workgroups fill the 160 KB of a CU -- 80 blocks per CU, all ~17 000 blocks of a piece resident at once)
				{ KernelTimer timer(ctx, "bgzf_inflate_tokens_kernel", (uint64_t) raw_size + stream_bytes, pieces);
				  #define TOKENS(LANES) bgzf_inflate_tokens_kernel<LANES><<<(n_blocks + LANES - 1) / LANES, LANES, 0, pieces>>>(ctx->ingest_raw[slot].as<uint8_t>(), ctx->ingest_blocks[slot].as<agpu_bgzf_block>(), n_blocks, target, \
				  	ctx->scratch("ingest.inflate_spill").as<uint8_t>(), notes.as<unsigned long long>(), note_count.as<uint32_t>(), status.as<int>(), failures)
				  if (lanes == 16) TOKENS(16); else if (lanes == 24) TOKENS(24); else TOKENS(20);
				  #undef TOKENS
				}
				bgzf_inflate_kernel<<<n_blocks, 64, 0, pieces>>>(ctx->ingest_raw[slot].as<uint8_t>(), ctx->ingest_blocks[slot].as<agpu_bgzf_block>(), target, ctx->scratch("ingest.inflate_spill").as<uint8_t>(), status.as<int>(), failures); // (returns at once but for the blocks handed back)
				{ KernelTimer timer(ctx, "bgzf_inflate_resolve_kernel", (uint64_t) stream_bytes, pieces);
				  bgzf_inflate_resolve_kernel<<<(n_blocks + BLOCK / 64 - 1) / (BLOCK / 64), BLOCK, 0, pieces>>>(ctx->ingest_blocks[slot].as<agpu_bgzf_block>(), n_blocks, target, ctx->scratch("ingest.inflate_spill").as<uint8_t>(),
				  	notes.as<unsigned long long>(), note_count.as<uint32_t>(), status.as<int>()); }
			}
		}
		else { KernelTimer timer(ctx, "bgzf_unwrap_kernel", (uint64_t) raw_size + stream_bytes, pieces);
		  bgzf_unwrap_kernel<<<n_blocks, BLOCK, 0, pieces>>>(ctx->ingest_raw[slot].as<uint8_t>(), ctx->ingest_blocks[slot].as<agpu_bgzf_block>(), ctx->ingest_stream.as<uint8_t>() + ctx->ingest_stream_size); }
		const uint64_t piece_stream_offset = ctx->ingest_stream_size;
		ctx->ingest_stream_size += stream_bytes;
		if (ctx->ingest_pushes > 0) HIP_CHECK(hipStreamWaitEvent(pieces, ctx->piece_ready[(ctx->ingest_pushes - 1) % AGPU_PIECE_SLOTS], 0)); // (ready in the order of the pieces, whichever stream they took)
		HIP_CHECK(hipEventRecord(ctx->piece_ready[slot], pieces));
		if (ctx->ingest_verify_crc) { // (~1 ms per 256 MB piece; between the copies on one stream it cost 0.3 s of a 54 GB file)
			KernelTimer timer(ctx, "bgzf_crc_kernel", deflated ? stream_bytes : raw_size, pieces);
			if (deflated) bgzf_crc_kernel<true><<<(n_blocks + 3) / 4, 256, 0, pieces>>>(ctx->ingest_stream.as<uint8_t>() + piece_stream_offset, ctx->ingest_blocks[slot].as<agpu_bgzf_block>(), n_blocks, ctx->scratch("ingest.crc_tables").as<Crc32Tables>(), ctx->scratch("ingest.crc_mismatches").as<unsigned int>());
			else bgzf_crc_kernel<false><<<(n_blocks + 3) / 4, 256, 0, pieces>>>(ctx->ingest_raw[slot].as<uint8_t>(), ctx->ingest_blocks[slot].as<agpu_bgzf_block>(), n_blocks, ctx->scratch("ingest.crc_tables").as<Crc32Tables>(), ctx->scratch("ingest.crc_mismatches").as<unsigned int>());
		}
		HIP_CHECK(hipEventRecord(ctx->piece_done[slot], pieces));
	} else { HIP_CHECK(hipEventRecord(ctx->piece_copied[slot], s)); HIP_CHECK(hipEventRecord(ctx->piece_ready[slot], s)); }
	TRY(wait_for_previous_push(ctx));
	++ctx->ingest_pushes;
	return windows_after_push(ctx);
}

int agpu_ingest_finish(agpu_ctx* ctx, agpu_ingest_result* result) {
	if (!ctx || !ctx->ingest_active) { set_last_error("agpu_ingest_begin must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	ctx->ingest_active = false;
	// until this function returns the stream and the tables of the ingest are in use: an allocation that fails in here may take back the scratch of the stages of the sample before
	// (DeviceBuffer::release_idle_buffers), never the buffers the pointers below point into
	struct Finishing { agpu_ctx* ctx; explicit Finishing(agpu_ctx* c) : ctx(c) { ctx->ingest_finishing = true; g_inside_ingest_finish = true; } ~Finishing() { ctx->ingest_finishing = false; g_inside_ingest_finish = false; } } finishing(ctx);
	const uint64_t size = ctx->ingest_stream_size, base = ctx->ingest_first_record;
	if (base > size) { set_last_error("failed to read SAM header"); return AGPU_ERR_INVALID; }
	if (agpu::debug_finish_runs_out_of_memory(ctx)) { set_last_error("hipMalloc failed (the buffers of the batch; asked for by agpu_debug_exhaust_memory_in_finish)"); return AGPU_ERR_NO_MEMORY; } // (test hook, include/arriba_gpu.h)
	if (!ctx->keeps_batch_buffers) take_sample_buffers(ctx); // (a session with two lanes: the batch of this sample is built where the sibling's last one, done on the device, lies -- unless the lanes keep their batch buffers, so that this ingest can be finished while the sibling's stages still run)
	HIP_CHECK(hipStreamSynchronize(ctx->piece_stream)); HIP_CHECK(hipStreamSynchronize(ctx->piece_stream2)); // (the last pieces unwrapped)
	if (ctx->ingest_verify_crc || ctx->ingest_deflated_pieces) {
		// a stored block whose payload does not give the CRC-32 of its trailer: the file is damaged (htslib: "CRC32 checksum mismatch").  A deflated block that did not decode (bad Huffman
		// code, output overrun, ISIZE mismatch) left its part of the stream unwritten: htslib fails on an inflate error whatever it does about checksums, so that counter is read
		// whenever a piece was deflated, also with ARRIBA_VERIFY_CRC=0 (advisor, round 4)
		unsigned int mismatches[2] = { 0, 0 };
		HIP_CHECK(hipMemcpyAsync(mismatches, ctx->scratch("ingest.crc_mismatches").ptr, 8, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		if (mismatches[1] > 0 || (ctx->ingest_verify_crc && mismatches[0] > 0)) { set_last_error("failed to load alignments"); return AGPU_ERR_INVALID; }
	}
	DeviceBuffer& counters = ctx->scratch("ingest.counters"); DeviceBuffer& rocprim_scratch = ctx->scratch("ingest.rocprim");
	ALLOC(counters, IC_COUNT * 4);
	uint32_t* device_counters = counters.as<uint32_t>();
	uint32_t host_counters[IC_COUNT];
	auto read_counters = [&]() -> int { HIP_CHECK(hipMemcpyAsync(host_counters, counters.ptr, IC_COUNT * 4, hipMemcpyDeviceToHost, s)); HIP_CHECK(hipStreamSynchronize(s)); return AGPU_OK; };
	(void) hipEventRecord(ctx->event_start, s);

	// 0. the windows that went through the front while the pieces arrived: the last one, then whether their runs are the groups of the names
	IngestProgress& progress = ctx->ingest_progress;
	bool streamed = false;
	if (progress.on) {
		if (!progress.abandoned) TRY(window_make(ctx, true));
		TRY(windows_advance(ctx, true));
		HIP_CHECK(hipStreamSynchronize(progress.work));
		for (size_t k = 0; k < progress.windows.size(); ++k) progress.events.push_back(progress.windows[k].readback);
		progress.windows.clear();
		streamed = !progress.abandoned;
		if (streamed) {
			TRY(read_counters());
			if (host_counters[IC_BROKEN] > 0 || host_counters[IC_COLLISION] != 0) streamed = false; // (a damaged record: the other way says so; two names with one key: hashed again there)
		}
		if (streamed && progress.heads > 1) { // a name with records in two places of the stream has two runs: the runs are not the reference's groups then
			DeviceBuffer& run_keys = ctx->scratch("ingest.run_keys"); DeviceBuffer& run_keys_sorted = ctx->scratch("ingest.run_keys_sorted");
			const uint32_t runs = (uint32_t) progress.heads;
			ALLOC(run_keys, (size_t) runs * 8); ALLOC(run_keys_sorted, (size_t) runs * 8);
			run_key_kernel<<<grid_for(runs), BLOCK, 0, s>>>(ctx->scratch("ingest.keys").as<uint64_t>(), ctx->scratch("ingest.group_first").as<uint32_t>(), runs, run_keys.as<uint64_t>());
			size_t temporary = 0;
			HIP_CHECK(rocprim::radix_sort_keys(nullptr, temporary, run_keys.as<uint64_t>(), run_keys_sorted.as<uint64_t>(), runs, 0, 64, s));
			if (temporary > rocprim_scratch.capacity) ALLOC(rocprim_scratch, temporary);
			{ KernelTimer timer(ctx, "rocprim::radix_sort_keys(keys of the runs)", (uint64_t) runs * 16);
			  HIP_CHECK(rocprim::radix_sort_keys(rocprim_scratch.ptr, temporary, run_keys.as<uint64_t>(), run_keys_sorted.as<uint64_t>(), runs, 0, 64, s)); }
			HIP_CHECK(hipMemsetAsync(device_counters + IC_MISMATCH, 0, 4, s));
			run_repeat_kernel<<<tally_grid(runs, BLOCK), BLOCK, 0, s>>>(run_keys_sorted.as<uint64_t>(), runs, device_counters + IC_MISMATCH);
			TRY(read_counters());
			if (host_counters[IC_MISMATCH] != 0) streamed = false;
		}
		if (!streamed && progress.touched) { // what the loop bodies of the windows have counted is counted again
			const uint64_t windows = ctx->host_coverage_window_offset.empty() ? 0 : ctx->host_coverage_window_offset.back();
			HIP_CHECK(hipMemsetAsync(ctx->coverage_windows32.ptr, 0, std::max<uint64_t>(coverage_difference_slots(windows, ctx->genome.n_contigs), 1) * 4, s));
			HIP_CHECK(hipMemsetAsync(ctx->coverage_fragment_starts.ptr, 0, std::max<uint64_t>(windows, 1), s));
			HIP_CHECK(hipMemsetAsync(ctx->coverage_fragment_ends.ptr, 0, std::max<uint64_t>(windows, 1), s));
			HIP_CHECK(hipMemsetAsync(ctx->ingest_viral_counts.ptr, 0, std::max<size_t>(ctx->genome.n_contigs, 1) * 8, s));
		}
		progress.abandoned = !streamed;
	}
	const uint8_t* bytes = ctx->ingest_stream.as<uint8_t>();
	if (!streamed) HIP_CHECK(hipMemsetAsync(counters.ptr, 0, IC_COUNT * 4, s));

	// 1. the record chain
	const uint64_t n_segments = (size - base + SEGMENT_BYTES - 1) / SEGMENT_BYTES;
	DeviceBuffer& segment_first = ctx->scratch("ingest.segment_first"); DeviceBuffer& segment_end = ctx->scratch("ingest.segment_end"); DeviceBuffer& segment_end_before = ctx->scratch("ingest.segment_end_before");
	DeviceBuffer& segment_count = ctx->scratch("ingest.segment_count"); DeviceBuffer& segment_base = ctx->scratch("ingest.segment_base"); DeviceBuffer& segment_mismatch = ctx->scratch("ingest.segment_mismatch");
	DeviceBuffer& record_offset = ctx->scratch("ingest.record_offset");
	uint64_t n_records = streamed ? progress.records : 0;
	if (n_segments > 0 && !streamed) {
		ALLOC(segment_first, n_segments * 8); ALLOC(segment_end, n_segments * 8); ALLOC(segment_end_before, n_segments * 8); ALLOC(segment_count, (n_segments + 1) * 4); ALLOC(segment_base, (n_segments + 1) * 4); ALLOC(segment_mismatch, n_segments);
		{ KernelTimer timer(ctx, "segment_guess_kernel", size - base);
		  segment_guess_kernel<<<grid_for(n_segments), BLOCK, 0, s>>>(bytes, size, base, 0, n_segments, ctx->ingest_n_targets, segment_first.as<uint64_t>(), segment_end.as<uint64_t>(), segment_count.as<uint32_t>()); }
		while (true) {
			HIP_CHECK(hipMemsetAsync(device_counters + IC_MISMATCH, 0, 4, s));
			segment_check_kernel<<<tally_grid(n_segments, BLOCK), BLOCK, 0, s>>>(segment_first.as<uint64_t>(), segment_end.as<uint64_t>(), 0, n_segments, segment_mismatch.as<uint8_t>(), device_counters + IC_MISMATCH);
			TRY(read_counters());
			if (host_counters[IC_MISMATCH] == 0) break;
			HIP_CHECK(hipMemcpyAsync(segment_end_before.ptr, segment_end.ptr, n_segments * 8, hipMemcpyDeviceToDevice, s));
			KernelTimer timer(ctx, "segment_repair_kernel", 0);
			segment_repair_kernel<<<grid_for(n_segments), BLOCK, 0, s>>>(bytes, size, base, 0, n_segments, segment_mismatch.as<uint8_t>(), segment_end_before.as<uint64_t>(), segment_first.as<uint64_t>(), segment_end.as<uint64_t>(), segment_count.as<uint32_t>());
		}
		uint64_t last_end = 0;
		HIP_CHECK(hipMemcpyAsync(&last_end, segment_end.as<uint64_t>() + (n_segments - 1), 8, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		if (last_end != size) { set_last_error("failed to load alignments"); return AGPU_ERR_INVALID; } // a record that does not end where the stream ends, or an impossible block size
		HIP_CHECK(hipMemsetAsync(segment_count.as<uint32_t>() + n_segments, 0, 4, s));
		size_t scan_bytes = 0;
		HIP_CHECK(rocprim::exclusive_scan(nullptr, scan_bytes, segment_count.as<uint32_t>(), segment_base.as<uint32_t>(), 0u, n_segments + 1, rocprim::plus<uint32_t>(), s));
		if (scan_bytes > rocprim_scratch.capacity) ALLOC(rocprim_scratch, scan_bytes);
		HIP_CHECK(rocprim::exclusive_scan(rocprim_scratch.ptr, scan_bytes, segment_count.as<uint32_t>(), segment_base.as<uint32_t>(), 0u, n_segments + 1, rocprim::plus<uint32_t>(), s));
		uint32_t total = 0;
		HIP_CHECK(hipMemcpyAsync(&total, segment_base.as<uint32_t>() + n_segments, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		n_records = total;
		// (the 32-bit scan wraps beyond 2^32 records: such a stream has more than 2^32 * 36 bytes)
		if ((size - base) / 36 >= 0xFFFFFFF0ull) { set_last_error("more than 2^32-16 alignment records: shard the input"); return AGPU_ERR_INVALID; }
		ALLOC(record_offset, std::max<uint64_t>(n_records, 1) * 8);
		{ KernelTimer timer(ctx, "segment_emit_kernel", n_records * 8);
		  segment_emit_kernel<<<grid_for(n_segments), BLOCK, 0, s>>>(bytes, size, base, 0, n_segments, segment_first.as<uint64_t>(), segment_base.as<uint32_t>(), record_offset.as<uint64_t>()); }
	}
	IngestStream in;
	in.bytes = bytes; in.size = size; in.record_offset = record_offset.as<uint64_t>(); in.n_records = n_records; in.n_targets = ctx->ingest_n_targets; in.tid_to_contig = ctx->ingest_tid_to_contig.as<uint32_t>();
	in.hit_index = nullptr; // (set where the records have been parsed: below)

	ctx->ingest_qname_runs = 0;
	if (ctx->ingest_part_of_sample && n_records > 0) { // the read names of this part, for the check that no name has records in another part (agpu_shard_merge)
		DeviceBuffer& run_flags = ctx->scratch("ingest.run_flags"); DeviceBuffer& run_starts = ctx->scratch("ingest.run_starts");
		ALLOC(run_flags, n_records); ALLOC(run_starts, n_records * 4);
		HIP_CHECK(hipMemsetAsync(device_counters, 0, IC_COUNT * 4, s));
		qname_run_flag_kernel<<<grid_for(n_records), BLOCK, 0, s>>>(in, n_records, run_flags.as<uint8_t>());
		TRY(select_flagged(ctx, rocprim_scratch, run_flags.as<uint8_t>(), run_starts.as<uint32_t>(), device_counters + IC_MAX_NAME, n_records));
		TRY(read_counters());
		const uint64_t n_runs = host_counters[IC_MAX_NAME];
		ALLOC(ctx->ingest_qname_keys, std::max<uint64_t>(n_runs, 1) * 16);
		if (n_runs > 0) qname_key_kernel<<<grid_for(n_runs), BLOCK, 0, s>>>(in, run_starts.as<uint32_t>(), n_runs, ctx->ingest_qname_keys.as<uint64_t>());
		ctx->ingest_qname_runs = n_runs;
		run_flags.release(); run_starts.release();
	}

	// 2. per record: status and name key; 3. records of one name adjacent
	DeviceBuffer& keys = ctx->scratch("ingest.keys"); DeviceBuffer& keys_sorted = ctx->scratch("ingest.keys_sorted"); DeviceBuffer& record_bits = ctx->scratch("ingest.record_bits");
	DeviceBuffer& sorted_records = ctx->scratch("ingest.sorted_records"); DeviceBuffer& head = ctx->scratch("ingest.head"); DeviceBuffer& group_start = ctx->scratch("ingest.group_start");
	DeviceBuffer& first_flags = ctx->scratch("ingest.first_flags"); DeviceBuffer& stream_rank = ctx->scratch("ingest.stream_rank");
	DeviceBuffer& group_first = ctx->scratch("ingest.group_first"); DeviceBuffer& group_begin = ctx->scratch("ingest.group_begin"); DeviceBuffer& group_count = ctx->scratch("ingest.group_count");
	const uint64_t records1 = std::max<uint64_t>(n_records, 1);
	if (!streamed) { ALLOC(keys, records1 * 8); ALLOC(keys_sorted, records1 * 8); ALLOC(record_bits, records1); ALLOC(sorted_records, records1 * 4); ALLOC(ctx->scratch("ingest.hit_index"), records1 * 4); }
	in.hit_index = ctx->scratch("ingest.hit_index").as<int32_t>(); // (of the windows, or filled by the pass below)
	uint64_t n_active = 0, seed = 0;
	uint32_t n_groups = 0;
	if (streamed) { // (the counters hold what the windows counted; the active records lie in sorted_records in the order of the stream, the runs are the groups)
		TRY(read_counters());
		n_active = progress.active; n_groups = (uint32_t) progress.heads;
		if (n_active != host_counters[IC_ACTIVE] || progress.groups_done != progress.heads) { set_last_error("the windows of the ingest lost count of their records"); return AGPU_ERR_DEVICE; }
		if (n_groups > 0) ALLOC(group_first, (size_t) n_groups * 4);
	}
	while (!streamed) {
		HIP_CHECK(hipMemsetAsync(device_counters, 0, IC_COUNT * 4, s));
		if (n_records > 0) {
			{ KernelTimer timer(ctx, "record_parse_kernel", n_records * (8 + 64 + 9));
			  record_parse_kernel<<<tally_grid(n_records, BLOCK) * 8, BLOCK, 0, s>>>(in, ctx->genome, seed, 0, keys.as<uint64_t>(), record_bits.as<uint8_t>(), ctx->scratch("ingest.hit_index").as<int32_t>(), device_counters); }
			TRY(sort_pairs<uint64_t>(ctx, rocprim_scratch, keys.as<uint64_t>(), keys_sorted.as<uint64_t>(), nullptr, sorted_records.as<uint32_t>(), n_records, 64, "rocprim::radix_sort_pairs(record keys)", true));
		}
		TRY(read_counters());
		if (host_counters[IC_BROKEN] > 0) { set_last_error("failed to load alignments"); return AGPU_ERR_INVALID; }
		n_active = host_counters[IC_ACTIVE];
		n_groups = 0;
		if (n_active > 0) {
			ALLOC(head, n_active); ALLOC(group_start, (n_active + 1) * 4);
			{ KernelTimer timer(ctx, "group_head_kernel", n_active * (8 + 1));
			  group_head_kernel<<<grid_for(n_active), BLOCK, 0, s>>>(keys_sorted.as<uint64_t>(), n_active, head.as<uint8_t>()); }
			TRY(select_flagged(ctx, rocprim_scratch, head.as<uint8_t>(), group_start.as<uint32_t>(), device_counters + IC_MAX_NAME /* borrowed as the count */, n_active));
			TRY(read_counters());
			n_groups = host_counters[IC_MAX_NAME];
			HIP_CHECK(hipMemsetAsync(device_counters + IC_MAX_NAME, 0, 4, s));
			// the groups in the order of their first records (see group_first_flag_kernel)
			ALLOC(first_flags, n_records); ALLOC(stream_rank, (n_records + 1) * 4); ALLOC(group_first, (size_t) n_groups * 4); ALLOC(group_begin, (size_t) n_groups * 4); ALLOC(group_count, (size_t) n_groups * 4);
			HIP_CHECK(hipMemsetAsync(first_flags.ptr, 0, n_records, s));
			group_first_flag_kernel<<<grid_for(n_groups), BLOCK, 0, s>>>(sorted_records.as<uint32_t>(), group_start.as<uint32_t>(), n_groups, first_flags.as<uint8_t>());
			{ size_t scan_bytes = 0;
			  HIP_CHECK(rocprim::exclusive_scan(nullptr, scan_bytes, first_flags.as<uint8_t>(), stream_rank.as<uint32_t>(), 0u, n_records, rocprim::plus<uint32_t>(), s));
			  if (scan_bytes > rocprim_scratch.capacity) ALLOC(rocprim_scratch, scan_bytes);
			  KernelTimer timer(ctx, "rocprim::exclusive_scan(first records)", n_records * 5);
			  HIP_CHECK(rocprim::exclusive_scan(rocprim_scratch.ptr, scan_bytes, first_flags.as<uint8_t>(), stream_rank.as<uint32_t>(), 0u, n_records, rocprim::plus<uint32_t>(), s)); }
			{ KernelTimer timer(ctx, "group_rank_kernel", (uint64_t) n_groups * 24);
			  group_rank_kernel<<<grid_for(n_groups), BLOCK, 0, s>>>(sorted_records.as<uint32_t>(), group_start.as<uint32_t>(), n_groups, n_active, stream_rank.as<uint32_t>(), group_first.as<uint32_t>(), group_begin.as<uint32_t>(), group_count.as<uint32_t>()); }
			{ KernelTimer timer(ctx, "group_names_kernel", n_active * (4 + 36 + 16));
			  group_names_kernel<<<grid_for(n_groups), BLOCK, 0, s>>>(in, sorted_records.as<uint32_t>(), group_begin.as<uint32_t>(), group_count.as<uint32_t>(), 0, n_groups, device_counters); }
			TRY(read_counters());
			if (host_counters[IC_COLLISION]) { seed += 0x632BE59BD9B4E019ull; continue; } // two names with one key: hash again with another seed (once in ~10^3 runs of 10^8 names)
		}
		break;
	}
	const uint64_t mapped_reads = host_counters[IC_MAPPED_READS], missing_hi_tag = host_counters[IC_MISSING_HI];

	// 4. the loop body of the reference per name; 5. the valid fragments in name order
	DeviceBuffer& plain = ctx->scratch("ingest.plain_plans"); DeviceBuffer& itd = ctx->scratch("ingest.itd_plans"); DeviceBuffer& valid = ctx->scratch("ingest.valid"); DeviceBuffer& sizes = ctx->scratch("ingest.sizes");
	DeviceBuffer& refs = ctx->scratch("ingest.refs"); DeviceBuffer& order = ctx->scratch("ingest.order"); DeviceBuffer& order_keys = ctx->scratch("ingest.order_keys"); DeviceBuffer& order_keys_sorted = ctx->scratch("ingest.order_keys_sorted");
	const uint64_t groups1 = std::max<uint32_t>(n_groups, 1);
	ALLOC(plain, groups1 * sizeof(FragmentPlan)); ALLOC(itd, groups1 * sizeof(TandemPlan)); ALLOC(valid, 2 * groups1); ALLOC(sizes, 2 * groups1 * sizeof(FragmentSizes)); // (behind the windows: large enough already, and filled)
	ALLOC(refs, 2 * groups1 * 4);
	uint64_t n_fragments = 0;
	if (n_groups > 0) {
		if (!streamed) { KernelTimer timer(ctx, "group_replay_kernel", size - base);
		  group_replay_kernel<false><<<grid_for(n_groups), BLOCK, 0, s>>>(replay_context(ctx, in), sorted_records.as<uint32_t>(), group_begin.as<uint32_t>(), group_count.as<uint32_t>(), 0, n_groups, plain.as<FragmentPlan>(), itd.as<TandemPlan>(), valid.as<uint8_t>(),
		                                                            sizes.as<FragmentSizes>(), ctx->ingest_viral_counts.as<unsigned long long>(), nullptr, device_counters); }
		TRY(select_flagged(ctx, rocprim_scratch, valid.as<uint8_t>(), refs.as<uint32_t>(), device_counters + IC_MAX_NAME, 2 * (uint64_t) n_groups));
		TRY(read_counters());
		n_fragments = host_counters[IC_MAX_NAME];
		HIP_CHECK(hipMemsetAsync(device_counters + IC_MAX_NAME, 0, 4, s));
	} else TRY(read_counters());
	const uint64_t malformed_count = host_counters[IC_MALFORMED];
	const bool no_chimeric_reads = host_counters[IC_CHIMERIC] == 0;
	if (n_fragments >= 0xFFFFFFF0ull) { set_last_error("a batch holds at most 2^32-16 fragments; shard larger inputs"); return AGPU_ERR_INVALID; }
	const uint64_t fragments1 = std::max<uint64_t>(n_fragments, 1);
	bool names_were_sorted = true;
	ALLOC(order, fragments1 * 4);
	if (n_fragments > 0) {
		ALLOC(order_keys, fragments1 * 8); ALLOC(order_keys_sorted, fragments1 * 8);
		HIP_CHECK(hipMemcpyAsync(order.ptr, refs.ptr, n_fragments * 4, hipMemcpyDeviceToDevice, s)); // ascending references == the order of first occurrence (the groups are numbered by their first records)
		const bool runs_in_name_order = streamed && host_counters[IC_RUNS_UNSORTED] == 0; // (told by the windows: group_replay_kernel<true>, run_itd_order_kernel)
		if (!runs_in_name_order) {
			KernelTimer timer(ctx, "name_order_check_kernel", n_fragments * (4 + 2 * 40));
			name_order_check_kernel<<<grid_for(n_fragments), BLOCK, 0, s>>>(in, group_first.as<uint32_t>(), order.as<uint32_t>(), n_fragments, device_counters);
		}
		TRY(read_counters());
		if (!runs_in_name_order && host_counters[IC_UNSORTED]) { // the names are not in std::string order: least-significant-chunk-first radix sort over the 8-byte chunks of the names
			names_were_sorted = false;
			const uint32_t chunks = (host_counters[IC_MAX_NAME] + 7) / 8;
			for (uint32_t chunk = chunks; chunk-- > 0; ) {
				{ KernelTimer timer(ctx, "name_chunk_kernel", n_fragments * (4 + 8 + 40));
				  name_chunk_kernel<<<grid_for(n_fragments), BLOCK, 0, s>>>(in, group_first.as<uint32_t>(), order.as<uint32_t>(), n_fragments, chunk, order_keys.as<uint64_t>()); }
				TRY(sort_pairs<uint64_t>(ctx, rocprim_scratch, order_keys.as<uint64_t>(), order_keys_sorted.as<uint64_t>(), order.as<uint32_t>(), refs.as<uint32_t>(), n_fragments, 64, "rocprim::radix_sort_pairs(name chunk)", false));
				order.swap(refs);
			}
		}
	}

	// 6. pool offsets, then the batch
	DeviceBuffer& cigar_words = ctx->scratch("ingest.cigar_words"); DeviceBuffer& sequence_bytes = ctx->scratch("ingest.sequence_bytes"); DeviceBuffer& name_lengths = ctx->scratch("ingest.name_lengths"); DeviceBuffer& new_group = ctx->scratch("ingest.new_group");
	DeviceBuffer& cigar_base = ctx->scratch("ingest.cigar_base"); DeviceBuffer& sequence_base = ctx->scratch("ingest.sequence_base"); DeviceBuffer& name_base = ctx->scratch("ingest.name_base"); DeviceBuffer& group_id = ctx->scratch("ingest.group_id");
	ALLOC(cigar_words, (n_fragments + 1) * 4); ALLOC(sequence_bytes, (n_fragments + 1) * 4); ALLOC(name_lengths, (n_fragments + 1) * 4); ALLOC(new_group, (n_fragments + 1) * 4);
	ALLOC(cigar_base, (n_fragments + 1) * 8); ALLOC(sequence_base, (n_fragments + 1) * 8); ALLOC(name_base, (n_fragments + 1) * 8); ALLOC(group_id, (n_fragments + 1) * 4);
	const uint32_t* qname_run = nullptr; // the QNAMEs of the runs numbered (equal numbers = equal QNAMEs), where the windows have told that this is so: see run_name_order_kernel
	if (streamed && names_were_sorted && n_fragments > 0 && n_groups > 0 && host_counters[IC_RUNS_UNSORTED] == 0 && host_counters[IC_QNAME_COMMA] == 0) {
		DeviceBuffer& qname_differs = ctx->scratch("ingest.qname_differs"); DeviceBuffer& numbered = ctx->scratch("ingest.qname_run");
		ALLOC(numbered, (size_t) n_groups * 4);
		size_t scan_bytes = 0;
		HIP_CHECK(rocprim::inclusive_scan(nullptr, scan_bytes, qname_differs.as<uint32_t>(), numbered.as<uint32_t>(), n_groups, rocprim::plus<uint32_t>(), s));
		if (scan_bytes > rocprim_scratch.capacity) ALLOC(rocprim_scratch, scan_bytes);
		HIP_CHECK(rocprim::inclusive_scan(rocprim_scratch.ptr, scan_bytes, qname_differs.as<uint32_t>(), numbered.as<uint32_t>(), n_groups, rocprim::plus<uint32_t>(), s));
		qname_run = numbered.as<uint32_t>();
	}
	{ KernelTimer timer(ctx, "fragment_layout_kernel", n_fragments * (4 + 8 + 16 + (qname_run ? 8 : 2 * 40)));
	  fragment_layout_kernel<<<grid_for(n_fragments + 1), BLOCK, 0, s>>>(in, group_first.as<uint32_t>(), order.as<uint32_t>(), n_fragments, sizes.as<FragmentSizes>(), qname_run,
	                                                                   cigar_words.as<uint32_t>(), sequence_bytes.as<uint32_t>(), name_lengths.as<uint32_t>(), new_group.as<uint32_t>()); }
	TRY(exclusive_sum_u64(ctx, rocprim_scratch, cigar_words.as<uint32_t>(), cigar_base.as<uint64_t>(), n_fragments + 1));
	TRY(exclusive_sum_u64(ctx, rocprim_scratch, sequence_bytes.as<uint32_t>(), sequence_base.as<uint64_t>(), n_fragments + 1));
	TRY(exclusive_sum_u64(ctx, rocprim_scratch, name_lengths.as<uint32_t>(), name_base.as<uint64_t>(), n_fragments + 1));
	if (n_fragments > 0) {
		size_t scan_bytes = 0;
		HIP_CHECK(rocprim::inclusive_scan(nullptr, scan_bytes, new_group.as<uint32_t>(), group_id.as<uint32_t>(), n_fragments, rocprim::plus<uint32_t>(), s));
		if (scan_bytes > rocprim_scratch.capacity) ALLOC(rocprim_scratch, scan_bytes);
		HIP_CHECK(rocprim::inclusive_scan(rocprim_scratch.ptr, scan_bytes, new_group.as<uint32_t>(), group_id.as<uint32_t>(), n_fragments, rocprim::plus<uint32_t>(), s));
	}
	uint64_t totals[3] = { 0, 0, 0 };
	HIP_CHECK(hipMemcpyAsync(&totals[0], cigar_base.as<uint64_t>() + n_fragments, 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipMemcpyAsync(&totals[1], sequence_base.as<uint64_t>() + n_fragments, 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipMemcpyAsync(&totals[2], name_base.as<uint64_t>() + n_fragments, 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	if (totals[0] >= 0xFFFFFFFFull || totals[1] / 4 >= 0xFFFFFFFFull) { set_last_error("batch too large for 32-bit pool offsets"); return AGPU_ERR_INVALID; } // (CIGAR words and sequence words: 16 GB each; the names have 64-bit offsets)
	const uint64_t n = n_fragments;
	ctx->n = n;
	ALLOC(ctx->n_aln, n); ALLOC(ctx->fbits, n); ALLOC(ctx->group, n * 4);
	for (int k = 0; k < 3; ++k) { ALLOC(ctx->contig[k], n * 2); ALLOC(ctx->start[k], n * 4); ALLOC(ctx->end[k], n * 4); ALLOC(ctx->abits[k], n); ALLOC(ctx->cigar_offset[k], n * 4); ALLOC(ctx->cigar_count[k], n * 2); }
	for (int k = 0; k < 2; ++k) { ALLOC(ctx->seq_offset[k], n * 4); ALLOC(ctx->seq_length[k], n * 4); }
	ALLOC(ctx->cigar_pool, totals[0] * 4); ALLOC(ctx->seq_pool, totals[1]); ALLOC(ctx->names, totals[2]); ALLOC(ctx->name_offset, (n + 1) * 8);
	ctx->names_size = totals[2]; ctx->ingest_pool_sizes[0] = totals[0]; ctx->ingest_pool_sizes[1] = totals[1];
	PackTarget target;
	fill_pack_target(ctx, target);
	{ KernelTimer timer(ctx, "fragment_pack_kernel", n * (1 + 1 + 4 + 3 * 17 + 16 + 4 + 24) + totals[0] * 8 + totals[1] * 2 + totals[2] * 2);
	  fragment_pack_kernel<<<grid_for(n + 1), BLOCK, 0, s>>>(in, group_first.as<uint32_t>(), order.as<uint32_t>(), n, plain.as<FragmentPlan>(), itd.as<TandemPlan>(),
	                                                       cigar_base.as<uint64_t>(), sequence_base.as<uint64_t>(), name_base.as<uint64_t>(), group_id.as<uint32_t>(), target, device_counters); }
	const uint64_t windows = ctx->host_coverage_window_offset.empty() ? 0 : ctx->host_coverage_window_offset.back();
	if (windows > 0) { // the difference array of the replay becomes the windows: one prefix sum over all contigs, then the counts to their places (32 bits for a part of a sample, 16 for the stages)
		const uint64_t slots = coverage_difference_slots(windows, ctx->genome.n_contigs);
		DeviceBuffer& summed = ctx->scratch("ingest.coverage_summed"); DeviceBuffer& rocprim_scratch = ctx->scratch("ingest.rocprim");
		ALLOC(summed, slots * 4);
		size_t temporary = 0;
		HIP_CHECK(rocprim::inclusive_scan(nullptr, temporary, ctx->coverage_windows32.as<uint32_t>(), summed.as<uint32_t>(), slots, rocprim::plus<uint32_t>(), s));
		ALLOC(rocprim_scratch, temporary);
		KernelTimer timer(ctx, "coverage_from_differences_kernel", slots * 8 + windows * 10);
		HIP_CHECK(rocprim::inclusive_scan(rocprim_scratch.ptr, temporary, ctx->coverage_windows32.as<uint32_t>(), summed.as<uint32_t>(), slots, rocprim::plus<uint32_t>(), s));
		coverage_from_differences_kernel<<<grid_for(windows), BLOCK, 0, s>>>(summed.as<uint32_t>(), ctx->coverage_window_offset.as<uint64_t>(), ctx->genome.n_contigs, windows, ctx->coverage_windows32.as<uint32_t>(), ctx->coverage_windows.as<uint16_t>());
	}
	TRY(read_counters());
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (size - base) * 2 + n_records * 40 + n * 250;
	ctx->max_read_length = host_counters[IC_MAX_READ_LENGTH];
	ctx->batch_input_bytes = n * (1 + 1 + 4 + 3 * 17 + 16) + totals[0] * 4 + totals[1];
	ctx->coverage.n_contigs = ctx->genome.n_contigs; ctx->coverage.window_offset = ctx->coverage_window_offset.as<uint64_t>(); ctx->coverage.coverage = ctx->coverage_windows.as<uint16_t>();
	ctx->coverage.fragment_starts = ctx->coverage_fragment_starts.as<uint8_t>(); ctx->coverage.fragment_ends = ctx->coverage_fragment_ends.as<uint8_t>();
	ctx->have_coverage = true;
	TRY(agpu::finish_batch_setup(ctx));
	ctx->batch_from_ingest = true;
	// The stream and the per-record tables are not needed any more, but they stay for the next sample: a resident service reads sample after sample, and mapping /
	// unmapping gigabytes costs as much as the kernels that use them (a 10^8-fragment sample: 54 GB of stream, 35 GB of tables; with the 48 GB of memo and task lists, the
	// read lists and the batch behind them ~190 GB of the 288 GB are in use).  Only when an allocation fails are they given back (DeviceBuffer::release_idle_buffers); a part
	// of a sample hands its windows on as they are (agpu_shard_export), everybody else is done with the 32-bit ones.
	if (getenv("ARRIBA_RELEASE_INGEST_BUFFERS") != nullptr) agpu::release_ingest_buffers(ctx); // (for measurements: the behaviour of round 2 on samples that leave less than 160 GB free)
	ctx->ingest_stream_size = 0;
	agpu_ingest_result& mine = ctx->ingest_result;
	memset(&mine, 0, sizeof(mine));
	mine.records = n_records; mine.fragments = n; mine.mapped_reads = mapped_reads; mine.malformed_count = malformed_count; mine.missing_hi_tag = missing_hi_tag;
	mine.no_chimeric_reads = no_chimeric_reads; mine.names_were_sorted = names_were_sorted; mine.stream_bytes = size; mine.windows = streamed ? (uint16_t) std::min<unsigned int>(progress.windows_made, 65535u) : 0;
	if (result) *result = mine;
	return AGPU_OK;
}

// ---- one sample over several GPUs: the batch of a part as one block of bytes, the blocks of all parts as the batch of the sample (shard_host.hpp) ------------

static int shard_header_of(agpu_ctx* ctx, ShardHeader& h) {
	memset(&h, 0, sizeof(h));
	const uint64_t n = ctx->n;
	h.magic = SHARD_MAGIC; h.n = n;
	uint32_t ends[4] = { 0, 0, 0, 0 }; // cigar words, sequence words, name bytes, last group id: what the last row says
	h.names_bytes = ctx->names_size;
	h.windows = ctx->host_coverage_window_offset.empty() ? 0 : ctx->host_coverage_window_offset.back(); h.n_contigs = ctx->genome.n_contigs;
	if (n > 0) {
		HIP_CHECK(hipMemcpy(&ends[3], ctx->group.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost));
		h.groups = (uint64_t) ends[3] + 1;
	}
	h.cigar_words = ctx->ingest_pool_sizes[0]; h.sequence_bytes = ctx->ingest_pool_sizes[1];
	const agpu_ingest_result& r = ctx->ingest_result;
	h.records = r.records; h.mapped_reads = r.mapped_reads; h.malformed_count = r.malformed_count; h.missing_hi_tag = r.missing_hi_tag; h.no_chimeric_reads = r.no_chimeric_reads;
	h.names_were_sorted = r.names_were_sorted; h.stream_bytes = r.stream_bytes; h.max_read_length = ctx->max_read_length;
	h.qname_runs = ctx->ingest_qname_runs;
	h.total_bytes = shard_layout(h).total;
	return AGPU_OK;
}

// the columns of the context's batch in the order of the sections (null where the section has no column of the context)
static void shard_columns(agpu_ctx* ctx, void* columns[SHARD_SECTIONS]) {
	for (int k = 0; k < SHARD_SECTIONS; ++k) columns[k] = nullptr;
	columns[SHARD_N_ALN] = ctx->n_aln.ptr; columns[SHARD_FBITS] = ctx->fbits.ptr; columns[SHARD_GROUP] = ctx->group.ptr;
	for (int slot = 0; slot < 3; ++slot) {
		void** c = columns + SHARD_SLOT0 + slot * SHARD_SLOT_FIELDS;
		c[SHARD_SLOT_CONTIG] = ctx->contig[slot].ptr; c[SHARD_SLOT_START] = ctx->start[slot].ptr; c[SHARD_SLOT_END] = ctx->end[slot].ptr; c[SHARD_SLOT_ABITS] = ctx->abits[slot].ptr;
		c[SHARD_SLOT_CIGAR_OFFSET] = ctx->cigar_offset[slot].ptr; c[SHARD_SLOT_CIGAR_COUNT] = ctx->cigar_count[slot].ptr;
	}
	columns[SHARD_SEQ_OFFSET0] = ctx->seq_offset[0].ptr; columns[SHARD_SEQ_LENGTH0] = ctx->seq_length[0].ptr; columns[SHARD_SEQ_OFFSET1] = ctx->seq_offset[1].ptr; columns[SHARD_SEQ_LENGTH1] = ctx->seq_length[1].ptr;
	columns[SHARD_CIGAR_POOL] = ctx->cigar_pool.ptr; columns[SHARD_SEQ_POOL] = ctx->seq_pool.ptr; columns[SHARD_NAME_OFFSET] = ctx->name_offset.ptr; columns[SHARD_NAMES] = ctx->names.ptr;
	columns[SHARD_WINDOWS32] = ctx->coverage_windows32.ptr; columns[SHARD_FRAGMENT_STARTS] = ctx->coverage_fragment_starts.ptr; columns[SHARD_FRAGMENT_ENDS] = ctx->coverage_fragment_ends.ptr;
	columns[SHARD_VIRAL_COUNTS] = ctx->ingest_viral_counts.ptr; columns[SHARD_QNAME_KEYS] = ctx->ingest_qname_keys.ptr;
}

int agpu_shard_export_size(agpu_ctx* ctx, uint64_t* bytes) {
	if (!ctx || !ctx->batch_from_ingest || !ctx->ingest_part_of_sample || ctx->annotated || !bytes) { set_last_error("agpu_shard_export works on the batch of an ingest with part_of_sample set, before any stage has run"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	ShardHeader h;
	TRY(shard_header_of(ctx, h));
	*bytes = h.total_bytes;
	return AGPU_OK;
}

int agpu_shard_export(agpu_ctx* ctx, void* block, uint64_t capacity) {
	if (!ctx || !ctx->batch_from_ingest || !ctx->ingest_part_of_sample || ctx->annotated || !block) { set_last_error("agpu_shard_export works on the batch of an ingest with part_of_sample set, before any stage has run"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	ShardHeader h;
	TRY(shard_header_of(ctx, h));
	if (capacity < h.total_bytes) { set_last_error("the block is smaller than agpu_shard_export_size says"); return AGPU_ERR_INVALID; }
	const ShardLayout layout = shard_layout(h);
	void* columns[SHARD_SECTIONS];
	shard_columns(ctx, columns);
	HIP_CHECK(hipMemcpyAsync(block, &h, sizeof(h), hipMemcpyDefault, s));
	for (int k = 0; k < SHARD_SECTIONS; ++k)
		if (layout.bytes[k] > 0) HIP_CHECK(hipMemcpyAsync((uint8_t*) block + layout.offset[k], columns[k], layout.bytes[k], hipMemcpyDefault, s));
	HIP_CHECK(hipStreamSynchronize(s));
	return AGPU_OK;
}

int agpu_shard_merge(agpu_ctx* ctx, const void* blocks, uint64_t stride, uint32_t n_parts, agpu_ingest_result* result) {
	if (!ctx || !blocks || n_parts == 0 || (stride & 15) != 0) { set_last_error("agpu_shard_merge takes the blocks of all parts at a stride that is a multiple of 16 bytes"); return AGPU_ERR_INVALID; }
	if (!ctx->have_annotation || !ctx->have_genome || ctx->host_coverage_window_offset.empty()) { set_last_error("the context must have ingested its own part before the parts are merged"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	{ // the kernels below read the blocks: blocks in host memory (a collective that went through the host) are brought to the device first
		hipPointerAttribute_t attributes;
		const bool on_device = hipPointerGetAttributes(&attributes, blocks) == hipSuccess && (attributes.type == hipMemoryTypeDevice || attributes.type == hipMemoryTypeManaged);
		(void) hipGetLastError();
		if (!on_device) {
			DeviceBuffer& staged = ctx->scratch("shard.blocks");
			ALLOC(staged, (size_t) n_parts * stride);
			HIP_CHECK(hipMemcpy(staged.ptr, blocks, (size_t) n_parts * stride, hipMemcpyHostToDevice));
			blocks = staged.ptr;
		}
	}
	std::vector<ShardHeader> parts(n_parts);
	for (uint32_t r = 0; r < n_parts; ++r) HIP_CHECK(hipMemcpy(&parts[r], (const uint8_t*) blocks + (size_t) r * stride, sizeof(ShardHeader), hipMemcpyDefault));
	ShardHeader total;
	const char* problem = shard_totals(parts.data(), n_parts, total);
	if (problem) { set_last_error(problem); return AGPU_ERR_INVALID; }
	for (uint32_t r = 0; r < n_parts; ++r) if (parts[r].total_bytes > stride) { set_last_error("a part of the sample is larger than the stride of the blocks"); return AGPU_ERR_INVALID; }
	if (total.windows != ctx->host_coverage_window_offset.back() || total.n_contigs != ctx->genome.n_contigs) { set_last_error("the parts of the sample were read against another assembly than this context holds"); return AGPU_ERR_INVALID; }
	// do the parts follow each other in the order of the names (the order of the reference's std::map)?  Every part is sorted, so its ends decide
	bool parts_in_name_order = true;
	{
		std::string previous; bool have_previous = false;
		for (uint32_t r = 0; r < n_parts; ++r) {
			if (parts[r].n == 0) continue;
			const ShardLayout layout = shard_layout(parts[r]);
			const uint8_t* block = (const uint8_t*) blocks + (size_t) r * stride;
			uint64_t offsets[2], last[2];
			HIP_CHECK(hipMemcpy(offsets, block + layout.offset[SHARD_NAME_OFFSET], 16, hipMemcpyDefault));
			HIP_CHECK(hipMemcpy(last, block + layout.offset[SHARD_NAME_OFFSET] + (parts[r].n - 1) * 8, 16, hipMemcpyDefault));
			if (offsets[1] < offsets[0] || last[1] < last[0] || last[1] > parts[r].names_bytes) { set_last_error("a part of the sample is damaged (name offsets)"); return AGPU_ERR_INVALID; }
			std::string first_name(offsets[1] - offsets[0], '\0'), last_name(last[1] - last[0], '\0');
			if (!first_name.empty()) HIP_CHECK(hipMemcpy(&first_name[0], block + layout.offset[SHARD_NAMES] + offsets[0], first_name.size(), hipMemcpyDefault));
			if (!last_name.empty()) HIP_CHECK(hipMemcpy(&last_name[0], block + layout.offset[SHARD_NAMES] + last[0], last_name.size(), hipMemcpyDefault));
			if (have_previous && !(previous < first_name)) parts_in_name_order = false; // (STAR writes the reads in the order of the FASTQ file: the merged batch is sorted below)
			previous = last_name; have_previous = true;
		}
	}
	if (total.qname_runs > 1) { // no read name in two parts (nor twice in one): the keys of all runs of names, sorted, neighbours compared
		const uint64_t runs = total.qname_runs;
		DeviceBuffer& all_keys = ctx->scratch("shard.qname_keys"); DeviceBuffer& low = ctx->scratch("shard.qname_low"); DeviceBuffer& low_sorted = ctx->scratch("shard.qname_low_sorted");
		DeviceBuffer& index_sorted = ctx->scratch("shard.qname_index"); DeviceBuffer& rocprim_scratch = ctx->scratch("shard.rocprim"); DeviceBuffer& small = ctx->scratch("shard.small");
		ALLOC(all_keys, runs * 16); ALLOC(low, runs * 8); ALLOC(low_sorted, runs * 8); ALLOC(index_sorted, runs * 4); ALLOC(small, 8);
		HIP_CHECK(hipMemsetAsync(small.ptr, 0, 8, s));
		uint64_t at = 0;
		for (uint32_t r = 0; r < n_parts; ++r) {
			if (parts[r].qname_runs == 0) continue;
			HIP_CHECK(hipMemcpyAsync(all_keys.as<uint64_t>() + 2 * at, (const uint8_t*) blocks + (size_t) r * stride + shard_layout(parts[r]).offset[SHARD_QNAME_KEYS], parts[r].qname_runs * 16, hipMemcpyDefault, s));
			at += parts[r].qname_runs;
		}
		qname_low_kernel<<<grid_for(runs), BLOCK, 0, s>>>(all_keys.as<uint64_t>(), runs, low.as<uint64_t>());
		TRY(sort_pairs<uint64_t>(ctx, rocprim_scratch, low.as<uint64_t>(), low_sorted.as<uint64_t>(), nullptr, index_sorted.as<uint32_t>(), runs, 64, "rocprim::radix_sort_pairs(read names of the parts)", true));
		qname_equal_kernel<<<grid_for(runs), BLOCK, 0, s>>>(low_sorted.as<uint64_t>(), index_sorted.as<uint32_t>(), all_keys.as<uint64_t>(), runs, small.as<uint32_t>());
		uint32_t equal = 0;
		HIP_CHECK(hipMemcpyAsync(&equal, small.ptr, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		all_keys.release(); low.release(); low_sorted.release(); index_sorted.release();
		if (equal > 0) { set_last_error("a read name occurs in more than one place of the sample (" + std::to_string(equal) + " times): the alignments of a read must follow each other in the file for it to be read in parts (STAR's output order; samtools collate otherwise)"); return AGPU_ERR_INVALID; }
	}
	const uint64_t n = total.n, windows = total.windows;
	ctx->have_batch = false; ctx->batch_from_ingest = false;
	ctx->n = n;
	ALLOC(ctx->n_aln, n); ALLOC(ctx->fbits, n); ALLOC(ctx->group, n * 4);
	for (int k = 0; k < 3; ++k) { ALLOC(ctx->contig[k], n * 2); ALLOC(ctx->start[k], n * 4); ALLOC(ctx->end[k], n * 4); ALLOC(ctx->abits[k], n); ALLOC(ctx->cigar_offset[k], n * 4); ALLOC(ctx->cigar_count[k], n * 2); }
	for (int k = 0; k < 2; ++k) { ALLOC(ctx->seq_offset[k], n * 4); ALLOC(ctx->seq_length[k], n * 4); }
	ALLOC(ctx->cigar_pool, total.cigar_words * 4); ALLOC(ctx->seq_pool, total.sequence_bytes); ALLOC(ctx->names, total.names_bytes); ALLOC(ctx->name_offset, (n + 1) * 8);
	ALLOC(ctx->coverage_windows32, std::max<uint64_t>(windows, 1) * 4);
	HIP_CHECK(hipMemsetAsync(ctx->coverage_windows32.ptr, 0, std::max<uint64_t>(windows, 1) * 4, s));
	HIP_CHECK(hipMemsetAsync(ctx->coverage_fragment_starts.ptr, 0, std::max<uint64_t>(windows, 1), s));
	HIP_CHECK(hipMemsetAsync(ctx->coverage_fragment_ends.ptr, 0, std::max<uint64_t>(windows, 1), s));
	HIP_CHECK(hipMemsetAsync(ctx->ingest_viral_counts.ptr, 0, std::max<size_t>(total.n_contigs, 1) * 8, s));
	void* columns[SHARD_SECTIONS];
	shard_columns(ctx, columns);
	static const uint8_t element_bytes[SHARD_SECTIONS] = { 1, 1, 4, 2, 4, 4, 1, 4, 2, 2, 4, 4, 1, 4, 2, 2, 4, 4, 1, 4, 2, 4, 4, 4, 4, 4, 1, 8, 1, 4, 1, 1, 8 };
	(void) hipEventRecord(ctx->event_start, s);
	uint64_t row = 0, cigar_at = 0, sequence_at = 0, name_at = 0, group_at = 0;
	for (uint32_t r = 0; r < n_parts; ++r) {
		const ShardHeader& h = parts[r];
		const ShardLayout layout = shard_layout(h);
		const uint8_t* block = (const uint8_t*) blocks + (size_t) r * stride;
		const uint8_t* part_n_aln = block + layout.offset[SHARD_N_ALN];
		for (int k = 0; k < SHARD_WINDOWS32; ++k) {
			if (h.n == 0 && k != SHARD_NAME_OFFSET) continue;
			const void* in = block + layout.offset[k];
			uint64_t at = row; // rows of the parts before this one ...
			if (k == SHARD_CIGAR_POOL) at = cigar_at; else if (k == SHARD_SEQ_POOL) at = sequence_at; else if (k == SHARD_NAMES) at = name_at; // ... or what they put into the pool
			uint8_t* out = (uint8_t*) columns[k] + at * element_bytes[k];
			int slot = -1; uint32_t base = 0; uint64_t count = h.n;
			if (k == SHARD_GROUP) { slot = 3; base = (uint32_t) group_at; }
			else if (k == SHARD_NAME_OFFSET) { // (64-bit; the last part brings the end of the names)
				count = h.n + (r + 1 == n_parts ? 1 : 0);
				if (count > 0) shard_rebase_names_kernel<<<grid_for(count), BLOCK, 0, s>>>((uint64_t*) out, (const uint64_t*) in, count, name_at);
				continue;
			}
			else if (k == SHARD_SEQ_OFFSET0 || k == SHARD_SEQ_OFFSET1) { slot = k == SHARD_SEQ_OFFSET0 ? 0 : 1; base = (uint32_t) (sequence_at / 4); }
			else if (k >= SHARD_SLOT0 && k < SHARD_SEQ_OFFSET0 && (k - SHARD_SLOT0) % SHARD_SLOT_FIELDS == SHARD_SLOT_CIGAR_OFFSET) { slot = (k - SHARD_SLOT0) / SHARD_SLOT_FIELDS; base = (uint32_t) cigar_at; }
			if (slot >= 0) { if (count > 0) shard_rebase_kernel<<<grid_for(count), BLOCK, 0, s>>>((uint32_t*) out, (const uint32_t*) in, count, base, part_n_aln, (uint32_t) slot); }
			else if (layout.bytes[k] > 0) HIP_CHECK(hipMemcpyAsync(out, in, layout.bytes[k], hipMemcpyDefault, s));
		}
		if (windows > 0) shard_add_coverage_kernel<<<grid_for(windows), BLOCK, 0, s>>>(ctx->coverage_windows32.as<uint32_t>(), ctx->coverage_fragment_starts.as<uint8_t>(), ctx->coverage_fragment_ends.as<uint8_t>(),
			(const uint32_t*) (block + layout.offset[SHARD_WINDOWS32]), block + layout.offset[SHARD_FRAGMENT_STARTS], block + layout.offset[SHARD_FRAGMENT_ENDS], windows);
		if (total.n_contigs > 0) shard_add_counts_kernel<<<grid_for(total.n_contigs), BLOCK, 0, s>>>(ctx->ingest_viral_counts.as<unsigned long long>(), (const unsigned long long*) (block + layout.offset[SHARD_VIRAL_COUNTS]), (uint32_t) total.n_contigs);
		row += h.n; cigar_at += h.cigar_words; sequence_at += h.sequence_bytes; name_at += h.names_bytes; group_at += h.groups;
	}
	if (n == 0) HIP_CHECK(hipMemsetAsync(ctx->name_offset.ptr, 0, 8, s));
	if (windows > 0) coverage_clamp_kernel<<<grid_for(windows), BLOCK, 0, s>>>(ctx->coverage_windows32.as<uint32_t>(), windows, ctx->coverage_windows.as<uint16_t>());
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	ctx->last_bytes = 2 * (n * (1 + 1 + 4 + 3 * 17 + 16 + 4) + total.cigar_words * 4 + total.sequence_bytes + total.names_bytes) + (uint64_t) n_parts * windows * 6;
	ctx->names_size = total.names_bytes; ctx->max_read_length = (uint32_t) total.max_read_length;
	ctx->ingest_pool_sizes[0] = total.cigar_words; ctx->ingest_pool_sizes[1] = total.sequence_bytes;
	ctx->batch_input_bytes = n * (1 + 1 + 4 + 3 * 17 + 16) + total.cigar_words * 4 + total.sequence_bytes;
	ctx->coverage.n_contigs = ctx->genome.n_contigs; ctx->coverage.window_offset = ctx->coverage_window_offset.as<uint64_t>(); ctx->coverage.coverage = ctx->coverage_windows.as<uint16_t>();
	ctx->coverage.fragment_starts = ctx->coverage_fragment_starts.as<uint8_t>(); ctx->coverage.fragment_ends = ctx->coverage_fragment_ends.as<uint8_t>();
	ctx->have_coverage = true;
	TRY(agpu::finish_batch_setup(ctx));
	ctx->batch_from_ingest = true; ctx->ingest_part_of_sample = false;
	ctx->coverage_windows32.release(); ctx->scratch("shard.blocks").release();
	if (!parts_in_name_order && n > 1) {
		// The fragments of the whole sample in name order: least-significant-chunk-first radix sort over the 8-byte chunks of the packed names (as the ingest does
		// it for the records of one part), the rows moved by the kernels of agpu_gather_rows_*, the read-name groups numbered again.
		DeviceBuffer& order = ctx->scratch("shard.order"); DeviceBuffer& order_out = ctx->scratch("shard.order_out"); DeviceBuffer& keys = ctx->scratch("shard.keys"); DeviceBuffer& keys_out = ctx->scratch("shard.keys_out");
		DeviceBuffer& small = ctx->scratch("shard.small"); DeviceBuffer& rocprim_scratch = ctx->scratch("shard.rocprim"); DeviceBuffer& new_group = ctx->scratch("shard.new_group");
		ALLOC(order, n * 4); ALLOC(order_out, n * 4); ALLOC(keys, n * 8); ALLOC(keys_out, n * 8); ALLOC(small, 8); ALLOC(new_group, n * 4);
		HIP_CHECK(hipMemsetAsync(small.ptr, 0, 8, s));
		iota_kernel<<<grid_for(n), BLOCK, 0, s>>>(order.as<uint32_t>(), n);
		packed_name_length_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->name_offset.as<uint64_t>(), n, small.as<uint32_t>());
		uint32_t longest = 0;
		HIP_CHECK(hipMemcpyAsync(&longest, small.ptr, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		for (uint32_t chunk = (longest + 7) / 8; chunk-- > 0; ) {
			packed_name_chunk_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->names.as<char>(), ctx->name_offset.as<uint64_t>(), order.as<uint32_t>(), n, chunk, keys.as<uint64_t>());
			TRY(sort_pairs<uint64_t>(ctx, rocprim_scratch, keys.as<uint64_t>(), keys_out.as<uint64_t>(), order.as<uint32_t>(), order_out.as<uint32_t>(), n, 64, "rocprim::radix_sort_pairs(name chunk of the merged batch)", false));
			order.swap(order_out);
		}
		packed_name_neighbours_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->names.as<char>(), ctx->name_offset.as<uint64_t>(), order.as<uint32_t>(), n, new_group.as<uint32_t>(), small.as<uint32_t>() + 1);
		uint32_t equal_names = 0;
		HIP_CHECK(hipMemcpyAsync(&equal_names, small.as<uint32_t>() + 1, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		if (equal_names > 0) { ctx->have_batch = false; set_last_error("a read name occurs in more than one part of the sample: the alignments of a read must follow each other in the file (STAR's output order; samtools collate otherwise)"); return AGPU_ERR_INVALID; }
		uint64_t pool_sizes[3];
		TRY(gather_rows_prepare(ctx, order.as<uint32_t>(), n, pool_sizes));
		agpu_batch_rows nowhere; memset(&nowhere, 0, sizeof(nowhere)); // (the rows stay on the device: in the scratch buffers "gather.*")
		TRY(agpu_gather_rows_copy(ctx, &nowhere));
		{ // the gathered columns become the batch
			size_t bytes = 0;
			HIP_CHECK(rocprim::inclusive_scan(nullptr, bytes, new_group.as<uint32_t>(), ctx->scratch("gather.group").as<uint32_t>(), n, rocprim::plus<uint32_t>(), s));
			if (bytes > rocprim_scratch.capacity) ALLOC(rocprim_scratch, bytes);
			HIP_CHECK(rocprim::inclusive_scan(rocprim_scratch.ptr, bytes, new_group.as<uint32_t>(), ctx->scratch("gather.group").as<uint32_t>(), n, rocprim::plus<uint32_t>(), s));
			HIP_CHECK(hipStreamSynchronize(s));
			ctx->n_aln.swap(ctx->scratch("gather.n_aln")); ctx->fbits.swap(ctx->scratch("gather.fbits")); ctx->group.swap(ctx->scratch("gather.group"));
			static const char* const slot_names[3][6] = { { "gather.contig0", "gather.start0", "gather.end0", "gather.abits0", "gather.cigar_offset0", "gather.cigar_count0" }, { "gather.contig1", "gather.start1", "gather.end1", "gather.abits1", "gather.cigar_offset1", "gather.cigar_count1" },
			                                              { "gather.contig2", "gather.start2", "gather.end2", "gather.abits2", "gather.cigar_offset2", "gather.cigar_count2" } };
			for (int k = 0; k < 3; ++k) {
				ctx->contig[k].swap(ctx->scratch(slot_names[k][0])); ctx->start[k].swap(ctx->scratch(slot_names[k][1])); ctx->end[k].swap(ctx->scratch(slot_names[k][2])); ctx->abits[k].swap(ctx->scratch(slot_names[k][3]));
				ctx->cigar_offset[k].swap(ctx->scratch(slot_names[k][4])); ctx->cigar_count[k].swap(ctx->scratch(slot_names[k][5]));
			}
			ctx->seq_offset[0].swap(ctx->scratch("gather.seq_offset0")); ctx->seq_length[0].swap(ctx->scratch("gather.seq_length0")); ctx->seq_offset[1].swap(ctx->scratch("gather.seq_offset1")); ctx->seq_length[1].swap(ctx->scratch("gather.seq_length1"));
			ctx->cigar_pool.swap(ctx->scratch("gather.cigar_pool")); ctx->seq_pool.swap(ctx->scratch("gather.seq_pool")); ctx->name_offset.swap(ctx->scratch("gather.name_offset")); ctx->names.swap(ctx->scratch("gather.names"));
		}
		TRY(agpu::finish_batch_setup(ctx));
		ctx->batch_from_ingest = true;
		static const char* const temporary[] = { "shard.order", "shard.order_out", "shard.keys", "shard.keys_out", "shard.new_group", "shard.rocprim", "gather.n_aln", "gather.fbits", "gather.group", "gather.contig0", "gather.start0", "gather.end0",
			"gather.abits0", "gather.cigar_offset0", "gather.cigar_count0", "gather.contig1", "gather.start1", "gather.end1", "gather.abits1", "gather.cigar_offset1", "gather.cigar_count1", "gather.contig2", "gather.start2", "gather.end2",
			"gather.abits2", "gather.cigar_offset2", "gather.cigar_count2", "gather.seq_offset0", "gather.seq_length0", "gather.seq_offset1", "gather.seq_length1", "gather.cigar_pool", "gather.seq_pool", "gather.name_offset", "gather.names" };
		for (size_t k = 0; k < sizeof(temporary) / sizeof(temporary[0]); ++k) ctx->scratch(temporary[k]).release();
		total.names_were_sorted = 0;
	}
	agpu_ingest_result& whole = ctx->ingest_result;
	memset(&whole, 0, sizeof(whole));
	whole.records = total.records; whole.fragments = n; whole.mapped_reads = total.mapped_reads; whole.malformed_count = total.malformed_count; whole.missing_hi_tag = total.missing_hi_tag;
	whole.no_chimeric_reads = (uint8_t) total.no_chimeric_reads; whole.names_were_sorted = (uint8_t) total.names_were_sorted; whole.stream_bytes = total.stream_bytes;
	if (result) *result = whole;
	return AGPU_OK;
}

int agpu_get_viral_read_counts(agpu_ctx* ctx, uint64_t* counts) {
	if (!ctx || !ctx->batch_from_ingest || !counts) { set_last_error("no batch built by the ingest"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipMemcpy(counts, ctx->ingest_viral_counts.ptr, (size_t) ctx->genome.n_contigs * 8, hipMemcpyDeviceToHost));
	return AGPU_OK;
}

int agpu_get_coverage(agpu_ctx* ctx, uint16_t* coverage, uint8_t* fragment_starts, uint8_t* fragment_ends) {
	if (!ctx || !ctx->have_coverage) { set_last_error("no coverage on the device"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	const uint64_t windows = ctx->host_coverage_window_offset.empty() ? 0 : ctx->host_coverage_window_offset.back();
	if (windows == 0) return AGPU_OK;
	if (coverage) HIP_CHECK(hipMemcpy(coverage, ctx->coverage_windows.ptr, windows * 2, hipMemcpyDeviceToHost));
	if (fragment_starts) HIP_CHECK(hipMemcpy(fragment_starts, ctx->coverage_fragment_starts.ptr, windows, hipMemcpyDeviceToHost));
	if (fragment_ends) HIP_CHECK(hipMemcpy(fragment_ends, ctx->coverage_fragment_ends.ptr, windows, hipMemcpyDeviceToHost));
	return AGPU_OK;
}

// the votes of detect_strandedness (source/read_stats.cpp:94-143) among the first `wanted` informative fragments of this context, in name order
static int strandedness_votes(agpu_ctx* ctx, uint32_t wanted, uint32_t& informative, uint32_t& matching) {
	hipStream_t s = ctx->stream;
	DeviceBuffer& counters = ctx->scratch("strandedness.counters"); DeviceBuffer& flags = ctx->scratch("strandedness.flags");
	ALLOC(counters, IC_COUNT * 4);
	HIP_CHECK(hipMemsetAsync(counters.ptr, 0, IC_COUNT * 4, s));
	AnnotationView annotation = ctx->annotation; annotation.n_dummy = 0; // the GTF genes: the reference looks before the dummy genes exist (source/arriba.cpp:141-158)
	// the pristine alignment bits (the strands as ingested) are what the reference looks at, before assign_strands_from_strandedness
	BatchView batch = ctx->batch;
	for (int k = 0; k < 3; ++k) batch.abits[k] = ctx->pristine_abits[k].as<uint8_t>();
	uint32_t host_counters[IC_COUNT]; memset(host_counters, 0, sizeof(host_counters));
	uint64_t chunk = 1u << 20;
	for (uint64_t first = 0; first < ctx->n && host_counters[IC_STRAND_COUNT] < wanted; first += chunk, chunk *= 4) {
		const uint64_t count = std::min<uint64_t>(chunk, ctx->n - first);
		ALLOC(flags, count);
		strandedness_flag_kernel<<<grid_for(count), BLOCK, 0, s>>>(batch, annotation, first, count, flags.as<uint8_t>());
		strandedness_take_kernel<<<1, 64, 0, s>>>(flags.as<uint8_t>(), count, wanted, counters.as<uint32_t>());
		HIP_CHECK(hipMemcpyAsync(host_counters, counters.ptr, IC_COUNT * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
	}
	informative = host_counters[IC_STRAND_COUNT]; matching = host_counters[IC_STRAND_MATCHING];
	return AGPU_OK;
}
int agpu_detect_strandedness(agpu_ctx* ctx, int* strandedness) {
	if (!ctx || !ctx->have_batch || !ctx->have_annotation) { set_last_error("annotation and batch must be on the device"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	const uint32_t sample_size = 100;
	const float threshold = 0.95;
	uint32_t count = 0, matching = 0;
	TRY(strandedness_votes(ctx, sample_size, count, matching));
	int verdict = 0;
	if (count >= sample_size) {
		if (matching < (1 - threshold) * count) verdict = 2;
		else if (matching > threshold * count) verdict = 1;
	}
	if (strandedness) *strandedness = verdict;
	return AGPU_OK;
}
int agpu_strandedness_votes(agpu_ctx* ctx, uint32_t wanted, uint32_t* informative, uint32_t* matching) {
	if (!ctx || !ctx->have_batch || !ctx->have_annotation) { set_last_error("annotation and batch must be on the device"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	uint32_t count = 0, match = 0;
	if (wanted > 0) TRY(strandedness_votes(ctx, wanted, count, match));
	if (informative) *informative = count;
	if (matching) *matching = match;
	return AGPU_OK;
}

// ---- one sample over several GPUs, the reads sharded (include/arriba_gpu.h: "the READS SHARDED"; agpu_sharded.hip): this context keeps the fragments of its part ------------
int agpu_shard_boundary_names(agpu_ctx* ctx, char* first_name, char* last_name, uint32_t capacity) {
	if (!ctx || !ctx->batch_from_ingest || !first_name || !last_name || capacity == 0) { set_last_error("agpu_shard_boundary_names works on the batch of an ingest"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	first_name[0] = 0; last_name[0] = 0;
	const uint64_t n = ctx->n;
	if (n == 0) return AGPU_OK;
	uint64_t first[2], last[2];
	HIP_CHECK(hipMemcpy(first, ctx->name_offset.as<uint64_t>(), 16, hipMemcpyDeviceToHost));
	HIP_CHECK(hipMemcpy(last, ctx->name_offset.as<uint64_t>() + (n - 1), 16, hipMemcpyDeviceToHost));
	if (first[1] < first[0] || last[1] < last[0] || last[1] > ctx->names_size) { set_last_error("the names of the batch are damaged"); return AGPU_ERR_INVALID; }
	if (first[1] - first[0] + 1 > capacity || last[1] - last[0] + 1 > capacity) { set_last_error("a read name is longer than the buffer of agpu_shard_boundary_names"); return AGPU_ERR_INVALID; }
	if (first[1] > first[0]) HIP_CHECK(hipMemcpy(first_name, ctx->names.as<char>() + first[0], first[1] - first[0], hipMemcpyDeviceToHost));
	first_name[first[1] - first[0]] = 0;
	if (last[1] > last[0]) HIP_CHECK(hipMemcpy(last_name, ctx->names.as<char>() + last[0], last[1] - last[0], hipMemcpyDeviceToHost));
	last_name[last[1] - last[0]] = 0;
	return AGPU_OK;
}

int agpu_shard_keep(agpu_ctx* ctx, uint64_t first_rank, uint64_t global_n) {
	if (!ctx || !ctx->batch_from_ingest || !ctx->ingest_part_of_sample || ctx->annotated) { set_last_error("agpu_shard_keep works on the batch of an ingest with part_of_sample set, before any stage has run"); return AGPU_ERR_INVALID; }
	if (first_rank + ctx->n > global_n || global_n >= 0xFFFFFFF0ull) { set_last_error("shard range out of bounds"); return AGPU_ERR_INVALID; }
	ctx->batch.first_rank = first_rank; ctx->global_n = global_n;
	ctx->read_sharded = true; ctx->state_imported = false; ctx->sample_gene_read_counts_set = false;
	ctx->ingest_part_of_sample = false;
	ctx->coverage_windows32.release(); ctx->ingest_qname_keys.release(); // (what a part hands to agpu_shard_export: not needed, the parts stay where they are)
	return AGPU_OK;
}

__global__ void coverage_widen_kernel(const uint16_t* windows, uint64_t n, uint32_t* out) {
	const uint64_t w = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (w < n) out[w] = windows[w];
}
int agpu_coverage_partial(agpu_ctx* ctx, uint32_t* windows_out, uint8_t* fragment_starts, uint8_t* fragment_ends, uint64_t* viral_counts) {
	if (!ctx || !ctx->batch_from_ingest || !ctx->have_coverage) { set_last_error("no coverage on the device"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t windows = ctx->host_coverage_window_offset.empty() ? 0 : ctx->host_coverage_window_offset.back();
	if (windows > 0) {
		if (windows_out) { // (the 16-bit windows of the part, already saturated, as 32-bit words for the sum)
			DeviceBuffer& wide = ctx->scratch("sharded.coverage32");
			ALLOC(wide, windows * 4);
			coverage_widen_kernel<<<grid_for(windows), BLOCK, 0, s>>>(ctx->coverage_windows.as<uint16_t>(), windows, wide.as<uint32_t>());
			HIP_CHECK(hipMemcpyAsync(windows_out, wide.ptr, windows * 4, hipMemcpyDefault, s));
		}
		if (fragment_starts) HIP_CHECK(hipMemcpyAsync(fragment_starts, ctx->coverage_fragment_starts.ptr, windows, hipMemcpyDefault, s));
		if (fragment_ends) HIP_CHECK(hipMemcpyAsync(fragment_ends, ctx->coverage_fragment_ends.ptr, windows, hipMemcpyDefault, s));
	}
	if (viral_counts && ctx->genome.n_contigs > 0) HIP_CHECK(hipMemcpyAsync(viral_counts, ctx->ingest_viral_counts.ptr, (size_t) ctx->genome.n_contigs * 8, hipMemcpyDefault, s));
	HIP_CHECK(hipStreamSynchronize(s));
	return AGPU_OK;
}
__global__ void flags_to_bits_kernel(uint8_t* flags, uint64_t n) {
	const uint64_t w = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (w < n) flags[w] = flags[w] ? 1 : 0;
}
int agpu_coverage_total(agpu_ctx* ctx, const uint32_t* windows_in, const uint8_t* fragment_starts, const uint8_t* fragment_ends, const uint64_t* viral_counts) {
	if (!ctx || !ctx->batch_from_ingest || !ctx->have_coverage || !windows_in || !fragment_starts || !fragment_ends || !viral_counts) { set_last_error("no coverage on the device"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t windows = ctx->host_coverage_window_offset.empty() ? 0 : ctx->host_coverage_window_offset.back();
	if (windows > 0) {
		DeviceBuffer& wide = ctx->scratch("sharded.coverage32");
		ALLOC(wide, windows * 4);
		HIP_CHECK(hipMemcpyAsync(wide.ptr, windows_in, windows * 4, hipMemcpyDefault, s));
		coverage_clamp_kernel<<<grid_for(windows), BLOCK, 0, s>>>(wide.as<uint32_t>(), windows, ctx->coverage_windows.as<uint16_t>());
		HIP_CHECK(hipMemcpyAsync(ctx->coverage_fragment_starts.ptr, fragment_starts, windows, hipMemcpyDefault, s));
		HIP_CHECK(hipMemcpyAsync(ctx->coverage_fragment_ends.ptr, fragment_ends, windows, hipMemcpyDefault, s));
		flags_to_bits_kernel<<<grid_for(windows), BLOCK, 0, s>>>(ctx->coverage_fragment_starts.as<uint8_t>(), windows);
		flags_to_bits_kernel<<<grid_for(windows), BLOCK, 0, s>>>(ctx->coverage_fragment_ends.as<uint8_t>(), windows);
	}
	if (ctx->genome.n_contigs > 0) HIP_CHECK(hipMemcpyAsync(ctx->ingest_viral_counts.ptr, viral_counts, (size_t) ctx->genome.n_contigs * 8, hipMemcpyDefault, s));
	HIP_CHECK(hipStreamSynchronize(s));
	return AGPU_OK;
}

int agpu_get_read_lengths(agpu_ctx* ctx, uint64_t first, uint64_t count, uint32_t* mate1, uint32_t* mate2) {
	if (!ctx || !ctx->have_batch) { set_last_error("no batch on the device"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	if (first > ctx->n) first = ctx->n;
	if (count > ctx->n - first) count = ctx->n - first;
	if (count == 0) return AGPU_OK;
	if (mate1) HIP_CHECK(hipMemcpy(mate1, ctx->seq_length[0].as<uint32_t>() + first, count * 4, hipMemcpyDeviceToHost));
	if (mate2) HIP_CHECK(hipMemcpy(mate2, ctx->seq_length[1].as<uint32_t>() + first, count * 4, hipMemcpyDeviceToHost));
	return AGPU_OK;
}

int agpu_gather_rows_begin(agpu_ctx* ctx, const uint32_t* fragments, uint64_t n, uint64_t* cigar_pool_size, uint64_t* seq_pool_size, uint64_t* names_size) {
	if (!ctx || !ctx->have_batch) { set_last_error("no batch on the device"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	if (fragments != nullptr) for (uint64_t k = 0; k < n; ++k) if (fragments[k] >= ctx->n) { set_last_error("fragment index out of range"); return AGPU_ERR_INVALID; }
	uint64_t sizes[3];
	TRY(gather_rows_prepare(ctx, fragments, n, sizes));
	if (cigar_pool_size) *cigar_pool_size = sizes[0];
	if (seq_pool_size) *seq_pool_size = sizes[1];
	if (names_size) *names_size = sizes[2];
	return AGPU_OK;
}

// the rows `fragments` (host or device memory; null = all, in order) of the resident batch: sizes of their pools, where every row goes
static int gather_rows_prepare(agpu_ctx* ctx, const uint32_t* fragments, uint64_t n, uint64_t* pool_sizes) {
	hipStream_t s = ctx->stream;
	ctx->gather_all = fragments == nullptr;
	if (ctx->gather_all) n = ctx->n;
	else {
		ALLOC(ctx->gather_ids, std::max<uint64_t>(n, 1) * 4);
		if (n > 0) HIP_CHECK(hipMemcpyAsync(ctx->gather_ids.ptr, fragments, n * 4, hipMemcpyDefault, s));
	}
	ctx->gather_n = n;
	DeviceBuffer& cigar_words = ctx->scratch("gather.cigar_words"); DeviceBuffer& sequence_bytes = ctx->scratch("gather.sequence_bytes"); DeviceBuffer& name_lengths = ctx->scratch("gather.name_lengths"); DeviceBuffer& rocprim_scratch = ctx->scratch("gather.rocprim");
	ALLOC(cigar_words, (n + 1) * 4); ALLOC(sequence_bytes, (n + 1) * 4); ALLOC(name_lengths, (n + 1) * 4);
	ALLOC(ctx->gather_cigar_base, (n + 1) * 8); ALLOC(ctx->gather_seq_base, (n + 1) * 8); ALLOC(ctx->gather_name_base, (n + 1) * 8);
	const uint32_t* ids = ctx->gather_all ? nullptr : ctx->gather_ids.as<uint32_t>();
	gather_sizes_kernel<<<grid_for(n + 1), BLOCK, 0, s>>>(ctx->batch, ctx->batch_from_ingest ? ctx->name_offset.as<uint64_t>() : nullptr, ids, n, cigar_words.as<uint32_t>(), sequence_bytes.as<uint32_t>(), name_lengths.as<uint32_t>());
	TRY(exclusive_sum_u64(ctx, rocprim_scratch, cigar_words.as<uint32_t>(), ctx->gather_cigar_base.as<uint64_t>(), n + 1));
	TRY(exclusive_sum_u64(ctx, rocprim_scratch, sequence_bytes.as<uint32_t>(), ctx->gather_seq_base.as<uint64_t>(), n + 1));
	TRY(exclusive_sum_u64(ctx, rocprim_scratch, name_lengths.as<uint32_t>(), ctx->gather_name_base.as<uint64_t>(), n + 1));
	HIP_CHECK(hipMemcpyAsync(&ctx->gather_sizes[0], ctx->gather_cigar_base.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipMemcpyAsync(&ctx->gather_sizes[1], ctx->gather_seq_base.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipMemcpyAsync(&ctx->gather_sizes[2], ctx->gather_name_base.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	for (int k = 0; k < 3; ++k) pool_sizes[k] = ctx->gather_sizes[k];
	return AGPU_OK;
}

int agpu_gather_rows_copy(agpu_ctx* ctx, agpu_batch_rows* rows) {
	if (!ctx || !ctx->have_batch || !rows) { set_last_error("agpu_gather_rows_begin must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->gather_n, n1 = std::max<uint64_t>(n, 1);
	// staged in device buffers of the same layout, then copied column by column
	struct Column { const char* name; size_t bytes; void* host; size_t copy_bytes; };
	std::vector<Column> columns;
	columns.push_back({ "gather.n_aln", n1, rows->n_aln, n }); columns.push_back({ "gather.fbits", n1, rows->fbits, n }); columns.push_back({ "gather.group", n1 * 4, rows->group, n * 4 });
	static const char* const slot_names[3][6] = { { "gather.contig0", "gather.start0", "gather.end0", "gather.abits0", "gather.cigar_offset0", "gather.cigar_count0" }, { "gather.contig1", "gather.start1", "gather.end1", "gather.abits1", "gather.cigar_offset1", "gather.cigar_count1" },
	                                              { "gather.contig2", "gather.start2", "gather.end2", "gather.abits2", "gather.cigar_offset2", "gather.cigar_count2" } };
	for (int k = 0; k < 3; ++k) {
		columns.push_back({ slot_names[k][0], n1 * 2, rows->contig[k], n * 2 }); columns.push_back({ slot_names[k][1], n1 * 4, rows->start[k], n * 4 }); columns.push_back({ slot_names[k][2], n1 * 4, rows->end[k], n * 4 });
		columns.push_back({ slot_names[k][3], n1, rows->abits[k], n }); columns.push_back({ slot_names[k][4], n1 * 4, rows->cigar_offset[k], n * 4 }); columns.push_back({ slot_names[k][5], n1 * 2, rows->cigar_count[k], n * 2 });
	}
	columns.push_back({ "gather.seq_offset0", n1 * 4, rows->seq_offset[0], n * 4 }); columns.push_back({ "gather.seq_length0", n1 * 4, rows->seq_length[0], n * 4 });
	columns.push_back({ "gather.seq_offset1", n1 * 4, rows->seq_offset[1], n * 4 }); columns.push_back({ "gather.seq_length1", n1 * 4, rows->seq_length[1], n * 4 });
	columns.push_back({ "gather.cigar_pool", std::max<uint64_t>(ctx->gather_sizes[0], 1) * 4, rows->cigar_pool, ctx->gather_sizes[0] * 4 }); columns.push_back({ "gather.seq_pool", std::max<uint64_t>(ctx->gather_sizes[1], 4), rows->seq_pool, ctx->gather_sizes[1] });
	const bool rows_become_the_batch = rows->name_offset == nullptr && rows->names == nullptr; // (agpu_shard_merge sorting the merged batch: 64-bit name offsets, as every batch has them)
	if (!rows_become_the_batch && ctx->gather_sizes[2] >= 0xFFFFFFFFull) { set_last_error("the names of the rows asked for take 4 GB and more: fetch them in portions"); return AGPU_ERR_CAPACITY; }
	columns.push_back({ "gather.name_offset", (n + 1) * (rows_become_the_batch ? 8 : 4), rows->name_offset, (n + 1) * 4 }); columns.push_back({ "gather.names", std::max<uint64_t>(ctx->gather_sizes[2], 1), rows->names, ctx->gather_sizes[2] });
	for (size_t c = 0; c < columns.size(); ++c) ALLOC(ctx->scratch(columns[c].name), columns[c].bytes);
	PackTarget out;
	size_t c = 0;
	out.n_aln = ctx->scratch(columns[c++].name).as<uint8_t>(); out.fbits = ctx->scratch(columns[c++].name).as<uint8_t>(); out.group = ctx->scratch(columns[c++].name).as<uint32_t>();
	for (int k = 0; k < 3; ++k) {
		out.contig[k] = ctx->scratch(columns[c++].name).as<uint16_t>(); out.start[k] = ctx->scratch(columns[c++].name).as<int32_t>(); out.end[k] = ctx->scratch(columns[c++].name).as<int32_t>();
		out.abits[k] = ctx->scratch(columns[c++].name).as<uint8_t>(); out.cigar_offset[k] = ctx->scratch(columns[c++].name).as<uint32_t>(); out.cigar_count[k] = ctx->scratch(columns[c++].name).as<uint16_t>();
	}
	out.seq_offset[0] = ctx->scratch(columns[c++].name).as<uint32_t>(); out.seq_length[0] = ctx->scratch(columns[c++].name).as<uint32_t>();
	out.seq_offset[1] = ctx->scratch(columns[c++].name).as<uint32_t>(); out.seq_length[1] = ctx->scratch(columns[c++].name).as<uint32_t>();
	out.cigar_pool = ctx->scratch(columns[c++].name).as<uint32_t>(); out.seq_pool = ctx->scratch(columns[c++].name).as<uint8_t>();
	{ DeviceBuffer& offsets = ctx->scratch(columns[c++].name); out.name_offset = rows_become_the_batch ? offsets.as<uint64_t>() : nullptr; out.row_name_offset = rows_become_the_batch ? nullptr : offsets.as<uint32_t>(); }
	out.names = ctx->scratch(columns[c++].name).as<char>();
	DeviceBuffer& abits_pointers = ctx->scratch("gather.abits_pointers");
	ALLOC(abits_pointers, 3 * sizeof(void*));
	const uint8_t* pristine[3] = { ctx->pristine_abits[0].as<uint8_t>(), ctx->pristine_abits[1].as<uint8_t>(), ctx->pristine_abits[2].as<uint8_t>() };
	HIP_CHECK(hipMemcpyAsync(abits_pointers.ptr, pristine, sizeof(pristine), hipMemcpyHostToDevice, s));
	const uint32_t* ids = ctx->gather_all ? nullptr : ctx->gather_ids.as<uint32_t>();
	gather_copy_kernel<<<grid_for(n + 1), BLOCK, 0, s>>>(ctx->batch, ctx->pristine_fbits.as<uint8_t>(), abits_pointers.as<const uint8_t*>(), ctx->batch_from_ingest ? ctx->name_offset.as<uint64_t>() : nullptr,
	                                                     ctx->names.as<char>(), ids, n, ctx->gather_cigar_base.as<uint64_t>(), ctx->gather_seq_base.as<uint64_t>(), ctx->gather_name_base.as<uint64_t>(), out);
	for (size_t k = 0; k < columns.size(); ++k)
		if (columns[k].host != nullptr && columns[k].copy_bytes > 0) HIP_CHECK(hipMemcpyAsync(columns[k].host, ctx->scratch(columns[k].name).ptr, columns[k].copy_bytes, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	rows->n = n; rows->cigar_pool_size = ctx->gather_sizes[0]; rows->seq_pool_size = ctx->gather_sizes[1]; rows->names_size = ctx->gather_sizes[2];
	return AGPU_OK;
}

}
