// arriba_amd/csrc/device/annotate_core.hpp -- per-fragment annotation logic of the device path.
//
// One thread handles one chimeric fragment.  The functions restate, on the flattened interval index,
//   get_annotation_by_coordinate   reference: source/annotation.t.hpp:47-101
//   is_breakpoint_spliced          reference: source/annotation.cpp:379-429
//   annotate_alignment(s)          reference: source/annotation.cpp:431-555
//   strand assignment              reference: source/read_chimeric_alignments.cpp:775-790
//   gene fallback / dummy genes    reference: source/arriba.cpp:190-319
// They are __host__ __device__ so that tests/emu can single-step the identical code on a CPU-only box;
// the product only ever runs them inside the HIP kernels of annotate.hip.
#ifndef AGPU_ANNOTATE_CORE_HPP
#define AGPU_ANNOTATE_CORE_HPP 1

#include "views.hpp"

namespace agpu {

AGPU_HD uint32_t atomic_add_u32(uint32_t* address, uint32_t value) {
#if defined(__HIP_DEVICE_COMPILE__)
	return atomicAdd(address, value);
#else
	uint32_t old = *address; *address += value; return old;
#endif
}

// Small sorted set of ids.  The annotate and filter kernels wait on dependent gathers (SQ counters: ~75 % of the wave cycles parked on
// s_waitcnt), so what they need is wavefronts in flight, i.e. few registers.  A register file cannot be indexed with a run-time index:
// an indexed array goes to scratch memory (HBM -- 4 GB of scratch writes per annotate launch in the first version), while sixteen
// constant-indexed elements per set cost 90 VGPRs for the five sets of the annotation (3 waves per SIMD).  Almost every set holds one
// or two ids, so the set is split: the first SET_LOW elements live in registers (every access written out with constant indices, so
// that scalar replacement sees nothing but constants), the elements behind them in a small per-lane array in scratch memory that only
// the lanes with such a large set ever touch (a wavefront without one skips the block; copies move the tail only if there is one).
// The tail is a SEPARATE object (IdSetTail) that the set only points to: as a member array it would be indexed with a run-time index
// inside the same stack object as n and low[], and scalar replacement then leaves the whole object in memory (measured: 6 GB of
// scratch writes per annotate launch).  AGPU_IDSET(name) declares a set together with its tail.
const int SET_CAPACITY = 16;
const uint32_t SET_LOW = 4;
#define AGPU_EACH_LOW(F) F(0) F(1) F(2) F(3)
struct IdSetTail { uint32_t words[SET_CAPACITY - SET_LOW]; };
#define AGPU_IDSET(name) IdSetTail name##_tail; IdSet name(name##_tail.words)
#define AGPU_IDSET3(name) IdSetTail name##_tail[3]; IdSet name[3] = { IdSet(name##_tail[0].words), IdSet(name##_tail[1].words), IdSet(name##_tail[2].words) }
struct IdSet {
	uint32_t n;
	uint32_t overflow;
	uint32_t low[SET_LOW];
	uint32_t* high; // elements SET_LOW .. n-1, in the IdSetTail of this set
	AGPU_HD explicit IdSet(uint32_t* tail) : high(tail) {}
	IdSet(const IdSet&) = delete; // a copy would share the tail
	AGPU_HD IdSet& operator=(const IdSet& other) { copy_from(other); return *this; } // copies the elements, keeps its own tail
	AGPU_HD void copy_from(const IdSet& other) {
		n = other.n; overflow = other.overflow;
#define AGPU_COPY(j) low[j] = other.low[j];
		AGPU_EACH_LOW(AGPU_COPY)
#undef AGPU_COPY
		if (other.n > SET_LOW) for (uint32_t j = SET_LOW; j < other.n; ++j) high[j - SET_LOW] = other.high[j - SET_LOW];
	}
	AGPU_HD void clear() { n = 0; overflow = 0; }
	AGPU_HD uint32_t get(uint32_t k) const {
		if (k >= SET_LOW) return high[k - SET_LOW];
		uint32_t value = low[0];
#define AGPU_GET(j) value = (k == j##u) ? low[j] : value;
		AGPU_EACH_LOW(AGPU_GET)
#undef AGPU_GET
		return value;
	}
	AGPU_HD void put(uint32_t k, uint32_t x) {
		if (k >= SET_LOW) { high[k - SET_LOW] = x; return; }
#define AGPU_PUT(j) low[j] = (k == j##u) ? x : low[j];
		AGPU_EACH_LOW(AGPU_PUT)
#undef AGPU_PUT
	}
	AGPU_HD void push_back(uint32_t x) { // caller keeps the order
		if (n == SET_CAPACITY) { overflow = 1; return; }
		put(n, x);
		++n;
	}
	AGPU_HD bool contains(uint32_t x) const {
		bool present = false;
#define AGPU_CONTAINS(j) present = present || (j##u < n && low[j] == x);
		AGPU_EACH_LOW(AGPU_CONTAINS)
#undef AGPU_CONTAINS
		if (n > SET_LOW) for (uint32_t j = SET_LOW; j < n; ++j) present = present || high[j - SET_LOW] == x;
		return present;
	}
	AGPU_HD void insert(uint32_t x) {
		if (contains(x)) return;
		if (n == SET_CAPACITY) { overflow = 1; return; }
		uint32_t carry = x; // bubble x to its place: every larger element moves up by one
#define AGPU_INSERT(j) { const uint32_t current = low[j]; const bool exchange = j##u < n && current > carry; low[j] = (exchange || j##u == n) ? carry : current; carry = exchange ? current : carry; }
		AGPU_EACH_LOW(AGPU_INSERT)
#undef AGPU_INSERT
		if (n >= SET_LOW) {
			for (uint32_t j = SET_LOW; j < n; ++j) {
				const uint32_t current = high[j - SET_LOW];
				if (current > carry) { high[j - SET_LOW] = carry; carry = current; }
			}
			high[n - SET_LOW] = carry;
		}
		++n;
	}
	AGPU_HD void assign_single(uint32_t x) { n = 1; low[0] = x; }
};

AGPU_HD void intersect_sets(const IdSet& a, const IdSet& b, IdSet& out) {
	out.clear();
#define AGPU_INTERSECT(i) if (i##u < a.n && b.contains(a.low[i])) out.push_back(a.low[i]); /* a is ascending, so is the result */
	AGPU_EACH_LOW(AGPU_INTERSECT)
#undef AGPU_INTERSECT
	if (a.n > SET_LOW) for (uint32_t i = SET_LOW; i < a.n; ++i) { const uint32_t x = a.high[i - SET_LOW]; if (b.contains(x)) out.push_back(x); }
}
// intersection, or the union if the intersection is empty (reference: combine_annotations, source/annotation.t.hpp:47-53)
AGPU_HD void combine_sets(const IdSet& a, const IdSet& b, IdSet& out, bool make_union) {
	intersect_sets(a, b, out);
	if (out.n == 0 && make_union) {
		out = a;
#define AGPU_UNION(j) if (j##u < b.n) out.insert(b.low[j]);
		AGPU_EACH_LOW(AGPU_UNION)
#undef AGPU_UNION
		if (b.n > SET_LOW) for (uint32_t j = SET_LOW; j < b.n; ++j) out.insert(b.high[j - SET_LOW]);
		out.overflow |= a.overflow | b.overflow;
	}
}

AGPU_HD void load_genes(const BatchView& b, int slot, uint64_t i, IdSet& out) {
	out.clear();
	uint32_t count = b.gene_count[slot][i];
	const uint32_t* source = b.genes[slot] + i * GENE_INLINE;
	if (count > (uint32_t) GENE_INLINE) source = b.gene_pool + source[0];
	if (count > (uint32_t) SET_CAPACITY) { count = SET_CAPACITY; out.overflow = 1; }
#define AGPU_LOAD(k) if (k##u < count) out.low[k] = source[k];
	AGPU_EACH_LOW(AGPU_LOAD)
#undef AGPU_LOAD
	if (count > SET_LOW) for (uint32_t k = SET_LOW; k < count; ++k) out.high[k - SET_LOW] = source[k];
	out.n = count;
}

// returns false if the overflow pool is exhausted
AGPU_HD bool store_genes(const BatchView& b, int slot, uint64_t i, const IdSet& set) {
	uint32_t* inline_ids = b.genes[slot] + i * GENE_INLINE;
	b.gene_count[slot][i] = (uint8_t) set.n;
	if (set.n <= (uint32_t) GENE_INLINE) {
		if (set.n > 0) inline_ids[0] = set.low[0];
		if (set.n > 1) inline_ids[1] = set.low[1];
		return true;
	}
	uint32_t offset = atomic_add_u32(b.gene_pool_used, set.n);
	if (offset + set.n > b.gene_pool_capacity) {
		b.gene_count[slot][i] = 0;
		return false;
	}
	inline_ids[0] = offset;
#define AGPU_STORE(k) if (k##u < set.n) b.gene_pool[offset + k] = set.low[k];
	AGPU_EACH_LOW(AGPU_STORE)
#undef AGPU_STORE
	if (set.n > SET_LOW) for (uint32_t k = SET_LOW; k < set.n; ++k) b.gene_pool[offset + k] = set.high[k - SET_LOW];
	return true;
}

// ---- flattened interval index ---------------------------------------------------------------------

AGPU_HD uint32_t index_lower_bound(const FlatIndexView& index, uint32_t contig, int32_t position) {
	uint32_t lo = index.contig_offset[contig], hi = index.contig_offset[contig + 1];
	if (index.bins != nullptr) { // narrow [lo, hi] to the keys of one bin
		const uint32_t base = index.bins[contig], n_bins = index.bins[contig + 1] - base;
		if (position < 0) { if (n_bins > 0) hi = index.bins[base]; }
		else {
			const uint32_t bin = (uint32_t) position >> INDEX_BIN_SHIFT;
			if (bin + 1 >= n_bins) return hi; // behind the last key of the contig
			lo = index.bins[base + bin]; hi = index.bins[base + bin + 1];
		}
	}
	while (lo < hi) {
		uint32_t mid = lo + ((hi - lo) >> 1);
		if (index.keys[mid] < position) lo = mid + 1; else hi = mid;
	}
	return lo;
}

// The same result as index_lower_bound for a position that is expected close to key index `hint` (all look-ups of one alignment --
// its start, its end, its clip and intron positions -- land within a few keys of each other): gallop from the hint to bracket the
// answer, then bisect the bracket.  A binary search over a whole contig costs ~17 dependent loads, this one or two.
const uint32_t NO_HINT = 0xFFFFFFFFu;
AGPU_HD uint32_t index_lower_bound_near(const FlatIndexView& index, uint32_t contig, int32_t position, uint32_t hint) {
	const uint32_t begin = index.contig_offset[contig], end = index.contig_offset[contig + 1];
	if (hint == NO_HINT || hint < begin || hint > end) return index_lower_bound(index, contig, position);
	uint32_t lo, hi; // the answer lies in [lo, hi]
	if (hint < end && index.keys[hint] < position) {
		lo = hint + 1; hi = lo;
		uint32_t step = 1;
		while (hi < end && index.keys[hi] < position) { lo = hi + 1; hi = (end - hi > step) ? hi + step : end; step <<= 1; }
	} else {
		hi = hint; lo = hint;
		uint32_t step = 1;
		while (lo > begin && index.keys[lo - 1] >= position) { hi = lo - 1; lo = (lo - begin > step) ? lo - step : begin; step <<= 1; }
	}
	while (lo < hi) {
		uint32_t mid = lo + ((hi - lo) >> 1);
		if (index.keys[mid] < position) lo = mid + 1; else hi = mid;
	}
	return lo;
}

struct ListRef { const uint32_t* p; uint32_t n; };
AGPU_HD ListRef index_bucket(const FlatIndexView& index, uint32_t k) {
	ListRef list;
	uint32_t begin = index.member_offset[k];
	list.p = index.members + begin;
	list.n = index.member_offset[k + 1] - begin;
	return list;
}
AGPU_HD ListRef empty_list() { ListRef list; list.p = 0; list.n = 0; return list; }

struct IdentityMap { AGPU_HD uint32_t operator()(uint32_t id) const { return id; } };
struct ExonToGeneMap { const uint32_t* exon_gene; AGPU_HD uint32_t operator()(uint32_t exon) const { return exon_gene[exon]; } };

template <class Map> AGPU_HD uint32_t emit_intersection(ListRef a, ListRef b, const Map& map, IdSet& out) {
	uint32_t i = 0, j = 0, common = 0;
	while (i < a.n && j < b.n) {
		uint32_t x = a.p[i], y = b.p[j];
		if (x < y) ++i;
		else if (y < x) ++j;
		else { out.insert(map(x)); ++common; ++i; ++j; }
	}
	return common;
}
template <class Map> AGPU_HD void emit_all(ListRef a, const Map& map, IdSet& out) {
	for (uint32_t i = 0; i < a.n; ++i) out.insert(map(a.p[i]));
}

// features at [start,end]: (features at start) ∩ (features at end), or their union if that is empty; each side also
// takes the neighbouring bucket if its boundary lies within 2 bp (reference: source/annotation.t.hpp:55-101).
// The set algebra is done on feature ids; `map` translates the surviving ids (exon -> gene for the exon index).
// `hint` (in/out): a key index near the queried coordinates, NO_HINT if unknown; receives the index found for the start coordinate
template <class Map> AGPU_HD void query_by_coordinate(const FlatIndexView& index, uint32_t contig, int32_t start, int32_t end, const Map& map, IdSet& out, uint32_t& hint) {
	out.clear();
	if (contig >= index.n_contigs) return;
	uint32_t contig_begin = index.contig_offset[contig], contig_end = index.contig_offset[contig + 1];
	if (start == end) {
		uint32_t k = index_lower_bound_near(index, contig, start, hint);
		hint = k;
		if (k != contig_end) emit_all(index_bucket(index, k), map, out);
		return;
	}
	if (start > end) { int32_t t = start; start = end; end = t; }
	ListRef a = empty_list(), b = empty_list(), c = empty_list(), d = empty_list();
	uint32_t k = index_lower_bound_near(index, contig, start, hint);
	hint = k;
	if (k != contig_end) {
		a = index_bucket(index, k);
		if (index.keys[k] - start <= 2) {
			++k;
			if (k != contig_end) b = index_bucket(index, k);
		}
	}
	k = index_lower_bound_near(index, contig, end, hint);
	if (k != contig_end) c = index_bucket(index, k);
	if (k != contig_begin && contig_end > contig_begin) {
		--k;
		if (end - index.keys[k] <= 2) d = index_bucket(index, k);
	}
	// (a, c), (a, d), (b, c), (b, d); then a, b, c, d -- rolled loops, one instance of the merge in the code
	uint32_t common = 0;
	AGPU_NOUNROLL for (int pair = 0; pair < 4; ++pair) {
		const ListRef left = pair < 2 ? a : b, right = (pair & 1) ? d : c;
		common += emit_intersection(left, right, map, out);
	}
	if (common == 0) {
		AGPU_NOUNROLL for (int list = 0; list < 4; ++list) {
			const ListRef members = list == 0 ? a : list == 1 ? b : list == 2 ? c : d;
			emit_all(members, map, out);
		}
	}
}

template <class Map> AGPU_HD void query_by_coordinate(const FlatIndexView& index, uint32_t contig, int32_t start, int32_t end, const Map& map, IdSet& out) {
	uint32_t hint = NO_HINT;
	query_by_coordinate(index, contig, start, end, map, out, hint);
}

// reference: filter_exons_near_splice_site, source/annotation.cpp:379-401
AGPU_HD bool bucket_has_splice_site(const AnnotationView& ann, uint32_t gene, bool upstream, int32_t breakpoint, uint32_t k) {
	ListRef exons = index_bucket(ann.exon_index, k);
	for (uint32_t m = 0; m < exons.n; ++m) {
		uint32_t e = exons.p[m];
		if (ann.exon_gene[e] != gene) continue;
		int32_t previous = ann.exon_previous[e], next = ann.exon_next[e];
		if (upstream) {
			int32_t start = ann.exon_start[e];
			int32_t distance = start - breakpoint; if (distance < 0) distance = -distance;
			if (distance <= MAX_SPLICE_SITE_DISTANCE &&
			    (previous != -1 || (previous == -1 && next == -1 && ann.exon_cds_start[e] != -1) || start == ann.exon_cds_start[e]))
				return true;
		} else {
			int32_t end = ann.exon_end[e];
			int32_t distance = end - breakpoint; if (distance < 0) distance = -distance;
			if (distance <= MAX_SPLICE_SITE_DISTANCE &&
			    (next != -1 || (previous == -1 && next == -1 && ann.exon_cds_start[e] != -1) || end == ann.exon_cds_end[e]))
				return true;
		}
	}
	return false;
}

// reference: source/annotation.cpp:404-429
AGPU_HD bool is_breakpoint_spliced(const AnnotationView& ann, uint32_t gene, bool upstream, int32_t breakpoint, uint32_t hint = NO_HINT) {
	uint32_t contig = ann.gene_contig[gene];
	const FlatIndexView& index = ann.exon_index;
	if (contig >= index.n_contigs) return false;
	uint32_t contig_begin = index.contig_offset[contig], contig_end = index.contig_offset[contig + 1];
	if (contig_begin == contig_end) return false;
	const uint32_t at = index_lower_bound_near(index, contig, breakpoint, hint);
	// the buckets at, at + 1, at - 1 in the reference's order (a rolled loop: one instance of the bucket scan in the code)
	AGPU_NOUNROLL for (int probe = 0; probe < 3; ++probe) {
		const uint32_t k = probe == 0 ? at : probe == 1 ? at + 1 : at - 1;
		const bool exists = probe == 0 ? at != contig_end : probe == 1 ? (at != contig_end && at + 1 != contig_end) : at != contig_begin;
		if (exists && bucket_has_splice_site(ann, gene, upstream, breakpoint, k)) return true;
	}
	return false;
}

AGPU_HD bool complement_strand_if(bool strand, bool condition) { return condition ? !strand : strand; }

// reference: annotate_alignment, source/annotation.cpp:431-503.  `bits` carries strand / predicted strand in and out.
AGPU_HD void annotate_alignment(const BatchView& b, const AnnotationView& ann, uint64_t i, int slot, uint8_t& bits, IdSet& genes) {
	uint32_t contig = b.contig[slot][i];
	int32_t start = b.start[slot][i];
	ExonToGeneMap to_gene; to_gene.exon_gene = ann.exon_gene;
	uint32_t hint = NO_HINT; // key index of the alignment's start in the exon index: every later look-up of this alignment is close to it
	query_by_coordinate(ann.exon_index, contig, start, b.end[slot][i], to_gene, genes, hint);

	uint32_t n_cigar = b.cigar_count[slot][i];
	bool ambiguous = bits & ABIT_PREDICTED_STRAND_AMBIGUOUS;
	if (n_cigar > 1 && (genes.n > 1 || ambiguous)) {
		const uint32_t* cigar = b.cigar_pool + b.cigar_offset[slot][i];
		AGPU_IDSET(supported); supported.clear();
		int32_t reference_position = start;
		for (uint32_t c = 0; c < n_cigar && supported.n == 0; ++c) {
			uint32_t op = cigar[c] & 15, length = cigar[c] >> 4;
			bool is_clip = (op == CIGAR_S || op == CIGAR_H);
			if (is_clip || op == CIGAR_N) {
				for (uint32_t g = 0; g < genes.n; ++g) {
					uint32_t gene = genes.get(g);
					// clip: the clipped end must be a splice site (upstream for the first CIGAR operation); intron: either end must be one
					// (the genes of an alignment lie on its contig, so the hint is a key index of the right contig)
					bool spliced = false;
					AGPU_NOUNROLL for (int end = 0; end < (is_clip ? 1 : 2) && !spliced; ++end) {
						const bool upstream = is_clip ? c == 0 : end == 1;
						spliced = is_breakpoint_spliced(ann, gene, upstream, end == 1 ? reference_position + (int32_t) length : reference_position, hint);
					}
					if (spliced) supported.push_back(gene);
				}
			}
			if (op == CIGAR_N || op == CIGAR_M || op == CIGAR_X || op == CIGAR_EQ || op == CIGAR_D)
				reference_position += length;
		}
		if (supported.n > 0) {
			if (supported.n < genes.n) genes = supported;
			if (ambiguous) {
				bool predicted = ann.gene_bits[supported.low[0]] & GBIT_STRAND;
				bool still_ambiguous = false;
				for (uint32_t g = 0; g < supported.n && !still_ambiguous; ++g)
					if (((ann.gene_bits[supported.get(g)] & GBIT_STRAND) != 0) != predicted)
						still_ambiguous = true;
				if (!still_ambiguous)
					bits = (bits & ~(ABIT_PREDICTED_STRAND | ABIT_PREDICTED_STRAND_AMBIGUOUS)) | (predicted ? ABIT_PREDICTED_STRAND : 0);
			}
		}
	}
}

AGPU_HD bool abit(uint8_t bits, uint8_t mask) { return (bits & mask) != 0; }
AGPU_HD void set_predicted(uint8_t& bits, bool strand) { bits = (bits & ~(ABIT_PREDICTED_STRAND | ABIT_PREDICTED_STRAND_AMBIGUOUS)) | (strand ? ABIT_PREDICTED_STRAND : 0); }
AGPU_HD void set_ambiguous(uint8_t& bits) { bits |= ABIT_PREDICTED_STRAND_AMBIGUOUS; }

AGPU_HD int32_t breakpoint_of(const BatchView& b, int slot, uint64_t i, uint8_t bits, bool split_read_convention) {
	// split read: forward -> start; supplementary / discordant mate: forward -> end
	bool forward = bits & ABIT_STRAND;
	if (split_read_convention) return forward ? b.start[slot][i] : b.end[slot][i];
	return forward ? b.end[slot][i] : b.start[slot][i];
}

// Stage 1 of the annotation of one fragment (reference: source/arriba.cpp:160-231):
// strands from strandedness, exon-based annotation, gene-index fallback, and the list of positions that map to no gene.
// unmapped_keys receives contig << 32 | position for every alignment end that needs a dummy gene.
// A fragment yields at most two such positions; they are returned in unmapped[0..n_unmapped) and appended to the global list by the caller
// (one atomic per workgroup on the device, see annotate_stage1_kernel).
AGPU_HD bool annotate_fragment_stage1(const BatchView& b, const AnnotationView& ann, uint32_t strandedness, uint64_t i, uint64_t unmapped[2], uint32_t& n_unmapped) {
	int n_aln = b.n_aln[i];
	uint8_t bits[3];
	AGPU_IDSET3(genes);
	// (the slot loops are unrolled so that bits[] and genes[] are only ever indexed with constants and stay in registers)
	AGPU_UNROLL for (int s = 0; s < 3; ++s) { bits[s] = (s < n_aln) ? (uint8_t) (b.abits[s][i] | ABIT_PREDICTED_STRAND_AMBIGUOUS) : 0; genes[s].clear(); }

	// reference: assign_strands_from_strandedness, source/read_chimeric_alignments.cpp:775-790
	if (strandedness != 0) {
		const bool mate1_is_first = abit(bits[MATE1], ABIT_FIRST_IN_PAIR);
		uint8_t first = mate1_is_first ? bits[MATE1] : bits[MATE2], second = mate1_is_first ? bits[MATE2] : bits[MATE1];
		bool first_predicted = complement_strand_if(abit(first, ABIT_STRAND), strandedness == 2);
		set_predicted(first, first_predicted);
		set_predicted(second, complement_strand_if(first_predicted, abit(first, ABIT_STRAND) == abit(second, ABIT_STRAND)));
		bits[MATE1] = mate1_is_first ? first : second; bits[MATE2] = mate1_is_first ? second : first;
		if (n_aln == 3)
			set_predicted(bits[SUPPLEMENTARY], complement_strand_if(abit(bits[SPLIT_READ], ABIT_PREDICTED_STRAND), abit(bits[SUPPLEMENTARY], ABIT_STRAND) != abit(bits[SPLIT_READ], ABIT_STRAND)));
	}

	// reference: annotate_alignments, source/annotation.cpp:505-555
	// One instance of annotate_alignment in the code, run n_aln times (a rolled loop over the uniform slot number): unrolled three times the
	// kernel was 44 k instructions, three times the instruction cache.  The results are copied to constant indices, so bits[] and genes[]
	// still live in registers.
	AGPU_NOUNROLL for (int s = 0; s < n_aln; ++s) {
		uint8_t bit = s == 0 ? bits[0] : s == 1 ? bits[1] : bits[2];
		AGPU_IDSET(found); found.clear();
		annotate_alignment(b, ann, i, s, bit, found);
		if (found.n > 0) bit |= ABIT_EXONIC;
		if (s == 0) { bits[0] = bit; genes[0] = found; } else if (s == 1) { bits[1] = bit; genes[1] = found; } else { bits[2] = bit; genes[2] = found; }
	}
	{
		bool amb1 = abit(bits[MATE1], ABIT_PREDICTED_STRAND_AMBIGUOUS), amb2 = abit(bits[MATE2], ABIT_PREDICTED_STRAND_AMBIGUOUS);
		bool same_strand = abit(bits[MATE1], ABIT_STRAND) == abit(bits[MATE2], ABIT_STRAND);
		if (amb1 && !amb2) set_predicted(bits[MATE1], complement_strand_if(abit(bits[MATE2], ABIT_PREDICTED_STRAND), same_strand));
		else if (!amb1 && amb2) set_predicted(bits[MATE2], complement_strand_if(abit(bits[MATE1], ABIT_PREDICTED_STRAND), same_strand));
		else if (!amb1 && !amb2) {
			if ((abit(bits[MATE1], ABIT_PREDICTED_STRAND) != abit(bits[MATE2], ABIT_PREDICTED_STRAND)) != same_strand) {
				set_ambiguous(bits[MATE1]); set_ambiguous(bits[MATE2]);
			}
		}
	}
	AGPU_IDSET(combined);
	if (n_aln == 3) {
		combine_sets(genes[SPLIT_READ], genes[MATE1], combined, true);
		if (genes[MATE1].n == 0 || combined.n < genes[MATE1].n) genes[MATE1] = combined;
		if (genes[SPLIT_READ].n == 0 || combined.n < genes[SPLIT_READ].n) genes[SPLIT_READ] = combined;
		bool amb_split = abit(bits[SPLIT_READ], ABIT_PREDICTED_STRAND_AMBIGUOUS), amb_supp = abit(bits[SUPPLEMENTARY], ABIT_PREDICTED_STRAND_AMBIGUOUS);
		bool different_strand = abit(bits[SUPPLEMENTARY], ABIT_STRAND) != abit(bits[SPLIT_READ], ABIT_STRAND);
		if (amb_split && !amb_supp) {
			bool predicted = complement_strand_if(abit(bits[SUPPLEMENTARY], ABIT_PREDICTED_STRAND), different_strand);
			set_predicted(bits[MATE1], predicted);
			set_predicted(bits[SPLIT_READ], predicted);
		} else if (!amb_split && amb_supp) {
			set_predicted(bits[SUPPLEMENTARY], complement_strand_if(abit(bits[SPLIT_READ], ABIT_PREDICTED_STRAND), different_strand));
		} else if (!amb_split && !amb_supp) {
			if ((abit(bits[SPLIT_READ], ABIT_PREDICTED_STRAND) != abit(bits[SUPPLEMENTARY], ABIT_PREDICTED_STRAND)) != different_strand) {
				set_ambiguous(bits[MATE1]); set_ambiguous(bits[SPLIT_READ]); set_ambiguous(bits[SUPPLEMENTARY]);
			}
		}
	}

	// reference: gene-index fallback, source/arriba.cpp:190-205
	IdentityMap identity;
	AGPU_NOUNROLL for (int s = 0; s < n_aln; ++s) {
		const uint32_t count = s == 0 ? genes[0].n : s == 1 ? genes[1].n : genes[2].n;
		if (count != 0) continue;
		AGPU_IDSET(found);
		query_by_coordinate(ann.gene_index, b.contig[s][i], b.start[s][i], b.end[s][i], identity, found);
		if (s == 0) genes[0] = found; else if (s == 1) genes[1] = found; else genes[2] = found;
	}
	if (n_aln == 3) {
		combine_sets(genes[SPLIT_READ], genes[MATE1], combined, true);
		if (genes[MATE1].n == 0 || combined.n < genes[MATE1].n) genes[MATE1] = combined;
		if (genes[SPLIT_READ].n == 0 || combined.n < genes[SPLIT_READ].n) genes[SPLIT_READ] = combined;
	}

	// reference: positions that need a dummy gene, source/arriba.cpp:207-231
	n_unmapped = 0;
	if (n_aln == 3) {
		if (genes[SPLIT_READ].n == 0)
			unmapped[n_unmapped++] = (uint64_t) b.contig[SPLIT_READ][i] << 32 | (uint32_t) breakpoint_of(b, SPLIT_READ, i, bits[SPLIT_READ], true);
		if (genes[SUPPLEMENTARY].n == 0)
			unmapped[n_unmapped++] = (uint64_t) b.contig[SUPPLEMENTARY][i] << 32 | (uint32_t) breakpoint_of(b, SUPPLEMENTARY, i, bits[SUPPLEMENTARY], false);
	} else {
		AGPU_UNROLL for (int s = 0; s < 2; ++s)
			if (genes[s].n == 0)
				unmapped[n_unmapped++] = (uint64_t) b.contig[s][i] << 32 | (uint32_t) breakpoint_of(b, s, i, bits[s], false);
	}

	bool ok = true;
	AGPU_UNROLL for (int s = 0; s < 3; ++s) {
		if (s >= n_aln) continue;
		b.abits[s][i] = bits[s];
		ok = store_genes(b, s, i, genes[s]) && !genes[s].overflow && ok;
	}
	return ok;
}

// ---- dummy genes ----------------------------------------------------------------------------------

AGPU_HD uint32_t lower_bound_u64(const uint64_t* keys, uint32_t n, uint64_t value) {
	uint32_t lo = 0, hi = n;
	while (lo < hi) {
		uint32_t mid = lo + ((hi - lo) >> 1);
		if (keys[mid] < value) lo = mid + 1; else hi = mid;
	}
	return lo;
}

// A new dummy gene starts at sorted position i (reference: source/arriba.cpp:243-259) when the contig changes, the gap to the
// previous position exceeds 10 kb, or a boundary key of the gene index lies in [previous position, this position].
AGPU_HD bool dummy_gene_starts_here(const uint64_t* sorted_keys, uint32_t i, const FlatIndexView& gene_index) {
	if (i == 0) return true;
	uint64_t previous = sorted_keys[i - 1], current = sorted_keys[i];
	uint32_t contig = (uint32_t) (current >> 32);
	if ((uint32_t) (previous >> 32) != contig) return true;
	int32_t previous_position = (int32_t) (uint32_t) previous, position = (int32_t) (uint32_t) current;
	if (previous_position + 10000 < position) return true;
	if (contig < gene_index.n_contigs) {
		uint32_t k = index_lower_bound(gene_index, contig, previous_position);
		if (k != gene_index.contig_offset[contig + 1] && gene_index.keys[k] <= position) return true;
	}
	return false;
}

// Result of a point query on the gene index as it looks after the dummy genes were added: GTF genes plus a contiguous
// range of dummy gene ids.  Many identical dummy genes can pile up on one position (every PCR duplicate of an intergenic
// read that sits exactly on a boundary key becomes its own dummy gene), so the range is kept symbolic.
struct GeneQuery {
	IdSet real;
	uint32_t dummy_first, dummy_count; // gene ids n_genes + dummy_first ... (all larger than any GTF gene id)
	AGPU_HD explicit GeneQuery(uint32_t* tail) : real(tail) {}
	AGPU_HD void clear() { real.clear(); dummy_first = 0; dummy_count = 0; }
	AGPU_HD uint32_t size() const { return real.n + dummy_count; }
	AGPU_HD uint32_t element(const AnnotationView& ann, uint32_t k) const { return k < real.n ? real.get(k) : ann.n_genes + dummy_first + (k - real.n); }
	AGPU_HD void assign_single(uint32_t gene) { real.clear(); real.assign_single(gene); dummy_first = 0; dummy_count = 0; }
	AGPU_HD void from_set(const IdSet& set) { real = set; dummy_first = 0; dummy_count = 0; }
	// false if the materialised set would not fit
	AGPU_HD bool to_set(const AnnotationView& ann, IdSet& out) const {
		out = real;
		for (uint32_t k = 0; k < dummy_count; ++k) out.insert(ann.n_genes + dummy_first + k);
		return !out.overflow;
	}
};

// Point query (reference: source/arriba.cpp:262-264 rebuilds the index with the dummy genes): the bucket of the first boundary
// key >= position, where the boundary keys of the dummy genes (end, start-1) are merged in.
AGPU_HD void query_point_with_dummy_genes(const AnnotationView& ann, uint32_t contig, int32_t position, GeneQuery& out) {
	out.clear();
	bool have_real = false, have_key = false;
	int32_t key = 0;
	uint32_t real_bucket = 0;
	if (contig < ann.gene_index.n_contigs) {
		uint32_t k = index_lower_bound(ann.gene_index, contig, position);
		if (k != ann.gene_index.contig_offset[contig + 1]) { have_real = true; real_bucket = k; key = ann.gene_index.keys[k]; have_key = true; }
	}
	if (ann.n_dummy > 0) {
		uint64_t base = (uint64_t) contig << 32;
		uint32_t j = lower_bound_u64(ann.dummy_end_key, ann.n_dummy, base | (uint32_t) position);
		if (j < ann.n_dummy && (uint32_t) (ann.dummy_end_key[j] >> 32) == contig) {
			int32_t candidate = (int32_t) (uint32_t) ann.dummy_end_key[j];
			if (!have_key || candidate < key) { key = candidate; have_key = true; }
		}
		j = lower_bound_u64(ann.dummy_start_key, ann.n_dummy, base | (uint32_t) (position + 1));
		if (j < ann.n_dummy && (uint32_t) (ann.dummy_start_key[j] >> 32) == contig) {
			int32_t candidate = (int32_t) (uint32_t) ann.dummy_start_key[j] - 1;
			if (!have_key || candidate < key) { key = candidate; have_key = true; }
		}
	}
	if (!have_key) return;
	if (have_real) { // the real genes containing `key` are those of the first real boundary >= position
		IdentityMap identity;
		emit_all(index_bucket(ann.gene_index, real_bucket), identity, out.real);
	}
	if (ann.n_dummy > 0) { // dummy genes containing `key`: a contiguous run (they are sorted and disjoint up to identical duplicates)
		uint64_t base = (uint64_t) contig << 32;
		uint32_t j = lower_bound_u64(ann.dummy_end_key, ann.n_dummy, base | (uint32_t) key);
		out.dummy_first = j;
		while (j < ann.n_dummy && (uint32_t) (ann.dummy_start_key[j] >> 32) == contig && (int32_t) (uint32_t) ann.dummy_start_key[j] <= key) ++j;
		out.dummy_count = j - out.dummy_first;
	}
}

AGPU_HD bool gene_contains(const AnnotationView& ann, uint32_t gene, int32_t position) { return ann.gene_start[gene] <= position && ann.gene_end[gene] >= position; }

// last element of the query that contains the position, or `fallback`
AGPU_HD uint32_t last_gene_containing(const AnnotationView& ann, const GeneQuery& genes, int32_t position, uint32_t fallback) {
	for (uint32_t k = genes.size(); k > 0; --k) {
		uint32_t gene = genes.element(ann, k - 1);
		if (gene_contains(ann, gene, position)) return gene;
	}
	return fallback;
}

// Stage 2 (reference: source/arriba.cpp:262-319): map still unannotated alignments to the dummy genes and reduce
// alignments that span several dummy genes to the one containing the breakpoint.
AGPU_HD bool annotate_fragment_stage2(const BatchView& b, const AnnotationView& ann, uint64_t i) {
	int n_aln = b.n_aln[i];
	IdSetTail genes_tail[3];
	GeneQuery genes[3] = { GeneQuery(genes_tail[0].words), GeneQuery(genes_tail[1].words), GeneQuery(genes_tail[2].words) };
	uint8_t bits[3];
	bool changed[3] = { false, false, false };
	AGPU_UNROLL for (int s = 0; s < 3; ++s) {
		genes[s].clear(); bits[s] = 0;
		if (s < n_aln) { AGPU_IDSET(loaded); load_genes(b, s, i, loaded); genes[s].from_set(loaded); bits[s] = b.abits[s][i]; }
	}

	if (n_aln == 3) {
		if (genes[MATE1].size() == 0 || genes[SPLIT_READ].size() == 0) {
			query_point_with_dummy_genes(ann, b.contig[SPLIT_READ][i], breakpoint_of(b, SPLIT_READ, i, bits[SPLIT_READ], true), genes[SPLIT_READ]);
			genes[MATE1] = genes[SPLIT_READ];
			changed[MATE1] = changed[SPLIT_READ] = true;
		}
		if (genes[SUPPLEMENTARY].size() == 0) {
			query_point_with_dummy_genes(ann, b.contig[SUPPLEMENTARY][i], breakpoint_of(b, SUPPLEMENTARY, i, bits[SUPPLEMENTARY], false), genes[SUPPLEMENTARY]);
			changed[SUPPLEMENTARY] = true;
		}
	} else {
		AGPU_UNROLL for (int s = 0; s < 2; ++s)
			if (genes[s].size() == 0) {
				query_point_with_dummy_genes(ann, b.contig[s][i], breakpoint_of(b, s, i, bits[s], false), genes[s]);
				changed[s] = true;
			}
	}

	AGPU_UNROLL for (int s = 0; s < 3; ++s) {
		if (s < n_aln && genes[s].size() > 1 && (ann.gene_bits[genes[s].element(ann, 0)] & GBIT_DUMMY)) {
			int32_t breakpoint = breakpoint_of(b, s, i, bits[s], true); // forward -> start for every slot here (source/arriba.cpp:291)
			uint32_t encompassing = last_gene_containing(ann, genes[s], breakpoint, genes[MATE1].element(ann, 0));
			genes[s].assign_single(encompassing);
			changed[s] = true;
		}
	}
	if (n_aln == 3 && genes[MATE1].size() > 0 && genes[SPLIT_READ].size() > 0) {
		uint32_t g1 = genes[MATE1].element(ann, 0), g2 = genes[SPLIT_READ].element(ann, 0);
		if (g1 != g2 && (ann.gene_bits[g1] & GBIT_DUMMY) && (ann.gene_bits[g2] & GBIT_DUMMY)) {
			int32_t breakpoint = breakpoint_of(b, SPLIT_READ, i, bits[SPLIT_READ], true);
			uint32_t encompassing = last_gene_containing(ann, genes[MATE1], breakpoint, g1);
			encompassing = last_gene_containing(ann, genes[SPLIT_READ], breakpoint, encompassing);
			genes[MATE1].assign_single(encompassing);
			genes[SPLIT_READ].assign_single(encompassing);
			changed[MATE1] = changed[SPLIT_READ] = true;
		}
	}
	bool ok = true;
	AGPU_UNROLL for (int s = 0; s < 3; ++s)
		if (s < n_aln && changed[s]) {
			AGPU_IDSET(out);
			ok = genes[s].to_set(ann, out) && ok;
			ok = store_genes(b, s, i, out) && ok;
		}
	return ok;
}

}

#endif
