/*
 * oracle/restate.cpp -- TEST INFRASTRUCTURE (the oracle), not product code.
 *
 * Independent CPU restatement of the reference's hot path for the checker role: it takes the same
 * structure-of-arrays views the device receives (include/arriba_gpu.h), rebuilds array-of-structs reads
 * and a std::map interval index from the raw gene/exon tables, and then walks the stages sequentially the
 * way the reference does -- no code is shared with arriba_amd/csrc/device.  Every function cites the
 * reference lines it follows.  It is pinned against the golden dumps of the real reference
 * (tests/golden/, produced by oracle/_ref/arriba_ref_dump) by tests/test_oracle_restatement.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>
#include "../include/arriba_gpu.h"

namespace {

typedef std::vector<int> gene_list; // sorted gene ids

struct Gene { int contig, start, end; bool forward, dummy; };
struct Exon { int start, end, gene, previous, next, cds_start, cds_end; };
struct Alignment {
	bool supplementary, first_in_pair, exonic, forward, predicted_forward, predicted_ambiguous;
	int contig, start, end;
	std::vector<uint32_t> cigar;
	std::string sequence;
	gene_list genes;
	int preclip() const { uint32_t op = cigar.front() & 15; return (op == 4 || op == 5) ? cigar.front() >> 4 : 0; }
	int postclip() const { uint32_t op = cigar.back() & 15; return (op == 4 || op == 5) ? cigar.back() >> 4 : 0; }
};
struct Read { std::vector<Alignment> a; bool single_end, multimapper, duplicate; int filter; uint32_t group; };
typedef std::map<int, std::vector<int> > bucket_map; // boundary key -> feature ids (ascending)

struct Candidate {
	int gene1, gene2, contig1, contig2, breakpoint1, breakpoint2;
	bool upstream1, upstream2, exonic1, exonic2, spliced1, spliced2, strand1, strand2, strands_ambiguous, start_gene1, start_ambiguous;
	int filter, split_reads1, split_reads2, discordant_mates, anchor1, anchor2;
	std::vector<int> list1, list2, discordant;
	Candidate(): exonic1(false), exonic2(false), spliced1(false), spliced2(false), strand1(true), strand2(true), strands_ambiguous(true), start_gene1(true), start_ambiguous(true),
	             filter(0), split_reads1(0), split_reads2(0), discordant_mates(0), anchor1(0), anchor2(0) {}
};

enum { F_NONE = 0, F_DUPLICATES = 1, F_INCONSISTENTLY_CLIPPED = 2, F_HOMOPOLYMER = 3, F_READ_THROUGH = 4, F_SAME_GENE = 5, F_SMALL_INSERT = 6, F_LONG_GAP = 7, F_HAIRPIN = 8,
       F_MISMATCHES = 10, F_UNINTERESTING = 30, F_VIRAL = 31, F_TOP_VIRAL = 32, F_LOW_COV_VIRAL = 33, F_LOW_ENTROPY = 36 };

}

struct oracle_state {
	agpu_params params;
	std::vector<Gene> genes;
	std::vector<Exon> exons;
	std::vector<bucket_map> exon_index, gene_index;
	std::vector<std::string> genome;
	std::vector<uint8_t> contig_bits;
	std::vector<Read> reads;
	size_t real_genes;
	std::vector<Candidate> candidates; // in order of first insertion
	std::vector<uint8_t> discordant_swapped;
	std::vector<uint64_t> remaining;
	std::string error;
};

namespace {

// reference: make_annotation_index, source/annotation.t.hpp:25-45 -- bucket(key) = features containing position `key`
template <class F> void build_index(const std::vector<F>& features, size_t count, size_t n_contigs, const std::vector<int>& contig_of, std::vector<bucket_map>& index) {
	index.assign(n_contigs, bucket_map());
	for (size_t f = 0; f < count; ++f) {
		bucket_map& map = index[contig_of[f]];
		bucket_map::iterator next = map.lower_bound(features[f].end);
		std::vector<int> copy = (next == map.end()) ? std::vector<int>() : next->second;
		if (map.find(features[f].end) == map.end()) map[features[f].end] = copy;
		next = map.lower_bound(features[f].start - 1);
		copy = (next == map.end()) ? std::vector<int>() : next->second;
		if (map.find(features[f].start - 1) == map.end()) map[features[f].start - 1] = copy;
		for (bucket_map::iterator b = map.lower_bound(features[f].end); b->first >= features[f].start; --b)
			b->second.insert(std::upper_bound(b->second.begin(), b->second.end(), (int) f), (int) f);
	}
}

gene_list set_union(const gene_list& a, const gene_list& b) { gene_list r; std::set_union(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(r)); return r; }
gene_list set_intersection(const gene_list& a, const gene_list& b) { gene_list r; std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(r)); return r; }
// reference: combine_annotations, source/annotation.t.hpp:47-53
gene_list combine(const gene_list& a, const gene_list& b, bool make_union = true) { gene_list r = set_intersection(a, b); if (r.empty() && make_union) r = set_union(a, b); return r; }

// reference: get_annotation_by_coordinate, source/annotation.t.hpp:55-101 (returns feature ids)
std::vector<int> features_at(const std::vector<bucket_map>& index, int contig, int start, int end) {
	std::vector<int> result;
	if (contig < 0 || (size_t) contig >= index.size()) return result;
	const bucket_map& map = index[contig];
	if (start == end) {
		bucket_map::const_iterator at = map.lower_bound(start);
		if (at != map.end()) result = at->second;
		return result;
	}
	if (start > end) std::swap(start, end);
	std::vector<int> at_start, at_end;
	bucket_map::const_iterator s = map.lower_bound(start);
	if (s != map.end()) {
		at_start = s->second;
		if (s->first - start <= 2) { ++s; if (s != map.end()) at_start = set_union(at_start, s->second); }
	}
	bucket_map::const_iterator e = map.lower_bound(end);
	if (e != map.end()) at_end = e->second;
	if (e != map.begin() && !map.empty()) { --e; if (end - e->first <= 2) at_end = set_union(at_end, e->second); }
	return combine(at_start, at_end);
}

// reference: filter_exons_near_splice_site + is_breakpoint_spliced, source/annotation.cpp:379-429
bool bucket_spliced(const oracle_state& o, int gene, bool upstream, int breakpoint, const std::vector<int>& exons) {
	for (size_t k = 0; k < exons.size(); ++k) {
		const Exon& x = o.exons[exons[k]];
		if (x.gene != gene) continue;
		bool single_coding = x.previous == -1 && x.next == -1 && x.cds_start != -1;
		if (upstream && std::abs(x.start - breakpoint) <= 2 && (x.previous != -1 || single_coding || x.start == x.cds_start)) return true;
		if (!upstream && std::abs(x.end - breakpoint) <= 2 && (x.next != -1 || single_coding || x.end == x.cds_end)) return true;
	}
	return false;
}
bool breakpoint_spliced(const oracle_state& o, int gene, bool upstream, int breakpoint) {
	int contig = o.genes[gene].contig;
	if ((size_t) contig >= o.exon_index.size() || o.exon_index[contig].empty()) return false;
	const bucket_map& map = o.exon_index[contig];
	bucket_map::const_iterator at = map.lower_bound(breakpoint);
	if (at != map.end()) {
		if (bucket_spliced(o, gene, upstream, breakpoint, at->second)) return true;
		bucket_map::const_iterator after = at; ++after;
		if (after != map.end() && bucket_spliced(o, gene, upstream, breakpoint, after->second)) return true;
	}
	if (at != map.begin()) { bucket_map::const_iterator before = at; --before; if (bucket_spliced(o, gene, upstream, breakpoint, before->second)) return true; }
	return false;
}

// reference: annotate_alignment, source/annotation.cpp:431-503
void annotate_alignment(const oracle_state& o, Alignment& a) {
	std::vector<int> exons = features_at(o.exon_index, a.contig, a.start, a.end);
	std::set<int> gene_set;
	for (size_t k = 0; k < exons.size(); ++k) gene_set.insert(o.exons[exons[k]].gene);
	a.genes.assign(gene_set.begin(), gene_set.end());
	if (a.cigar.size() > 1 && (a.genes.size() > 1 || a.predicted_ambiguous)) {
		gene_list supported;
		int position = a.start;
		for (size_t c = 0; c < a.cigar.size() && supported.empty(); ++c) {
			uint32_t op = a.cigar[c] & 15; int length = a.cigar[c] >> 4;
			if (op == 4 || op == 5 || op == 3) {
				for (size_t g = 0; g < a.genes.size(); ++g) {
					int gene = a.genes[g];
					bool drop;
					if (op == 3) drop = !breakpoint_spliced(o, gene, false, position) && !breakpoint_spliced(o, gene, true, position + length);
					else drop = (c == 0) ? !breakpoint_spliced(o, gene, true, position) : !breakpoint_spliced(o, gene, false, position);
					if (!drop) supported.push_back(gene);
				}
			}
			if (op == 3 || op == 0 || op == 8 || op == 7 || op == 2) position += length;
		}
		if (!supported.empty()) {
			if (supported.size() < a.genes.size()) a.genes = supported;
			if (a.predicted_ambiguous) {
				bool strand = o.genes[supported[0]].forward, mixed = false;
				for (size_t g = 0; g < supported.size(); ++g) if (o.genes[supported[g]].forward != strand) mixed = true;
				if (!mixed) { a.predicted_ambiguous = false; a.predicted_forward = strand; }
			}
		}
	}
}

bool flip_if(bool strand, bool condition) { return condition ? !strand : strand; }

// reference: assign_strands_from_strandedness (source/read_chimeric_alignments.cpp:775-790) + annotate_alignments (source/annotation.cpp:505-555)
void annotate_read(const oracle_state& o, Read& r) {
	std::vector<Alignment>& a = r.a;
	if (o.params.strandedness != 0) {
		int first = a[0].first_in_pair ? 0 : 1, second = 1 - first;
		a[first].predicted_forward = flip_if(a[first].forward, o.params.strandedness == 2); a[first].predicted_ambiguous = false;
		a[second].predicted_forward = flip_if(a[first].predicted_forward, a[first].forward == a[second].forward); a[second].predicted_ambiguous = false;
		if (a.size() == 3) { a[2].predicted_forward = flip_if(a[1].predicted_forward, a[2].forward != a[1].forward); a[2].predicted_ambiguous = false; }
	}
	for (size_t s = 0; s < a.size(); ++s) { annotate_alignment(o, a[s]); a[s].exonic = !a[s].genes.empty(); }
	if (a[0].predicted_ambiguous && !a[1].predicted_ambiguous) { a[0].predicted_forward = flip_if(a[1].predicted_forward, a[0].forward == a[1].forward); a[0].predicted_ambiguous = false; }
	else if (!a[0].predicted_ambiguous && a[1].predicted_ambiguous) { a[1].predicted_forward = flip_if(a[0].predicted_forward, a[0].forward == a[1].forward); a[1].predicted_ambiguous = false; }
	else if (!a[0].predicted_ambiguous && !a[1].predicted_ambiguous) {
		if ((a[0].predicted_forward != a[1].predicted_forward) != (a[0].forward == a[1].forward)) a[0].predicted_ambiguous = a[1].predicted_ambiguous = true;
	}
	if (a.size() == 3) {
		gene_list both = combine(a[1].genes, a[0].genes);
		if (a[0].genes.empty() || both.size() < a[0].genes.size()) a[0].genes = both;
		if (a[1].genes.empty() || both.size() < a[1].genes.size()) a[1].genes = both;
		if (a[1].predicted_ambiguous && !a[2].predicted_ambiguous) {
			a[0].predicted_forward = flip_if(a[2].predicted_forward, a[2].forward != a[1].forward); a[0].predicted_ambiguous = false;
			a[1].predicted_forward = a[0].predicted_forward; a[1].predicted_ambiguous = false;
		} else if (!a[1].predicted_ambiguous && a[2].predicted_ambiguous) {
			a[2].predicted_forward = flip_if(a[1].predicted_forward, a[2].forward != a[1].forward); a[2].predicted_ambiguous = false;
		} else if (!a[1].predicted_ambiguous && !a[2].predicted_ambiguous) {
			if ((a[1].predicted_forward != a[2].predicted_forward) != (a[1].forward != a[2].forward)) a[0].predicted_ambiguous = a[1].predicted_ambiguous = a[2].predicted_ambiguous = true;
		}
	}
}

int breakpoint_outer(const Alignment& a) { return a.forward ? a.end : a.start; }   // supplementary / discordant mate
int breakpoint_inner(const Alignment& a) { return a.forward ? a.start : a.end; }   // split read

// reference: source/arriba.cpp:190-325 -- gene fallback, dummy genes, multi-dummy resolution
void fallback_and_dummy_genes(oracle_state& o) {
	for (size_t i = 0; i < o.reads.size(); ++i) {
		std::vector<Alignment>& a = o.reads[i].a;
		for (size_t s = 0; s < a.size(); ++s) if (a[s].genes.empty()) a[s].genes = features_at(o.gene_index, a[s].contig, a[s].start, a[s].end);
		if (a.size() == 3) {
			gene_list both = combine(a[1].genes, a[0].genes);
			if (a[0].genes.empty() || both.size() < a[0].genes.size()) a[0].genes = both;
			if (a[1].genes.empty() || both.size() < a[1].genes.size()) a[1].genes = both;
		}
	}
	std::vector<std::pair<int,int> > unmapped; // (contig, position)
	for (size_t i = 0; i < o.reads.size(); ++i) {
		const std::vector<Alignment>& a = o.reads[i].a;
		if (a.size() == 3) {
			if (a[1].genes.empty()) unmapped.push_back(std::make_pair(a[1].contig, breakpoint_inner(a[1])));
			if (a[2].genes.empty()) unmapped.push_back(std::make_pair(a[2].contig, breakpoint_outer(a[2])));
		} else {
			for (size_t s = 0; s < a.size(); ++s) if (a[s].genes.empty()) unmapped.push_back(std::make_pair(a[s].contig, breakpoint_outer(a[s])));
		}
	}
	if (!unmapped.empty()) {
		std::stable_sort(unmapped.begin(), unmapped.end());
		Gene dummy; dummy.contig = unmapped[0].first; dummy.start = dummy.end = unmapped[0].second; dummy.forward = true; dummy.dummy = true;
		bucket_map::const_iterator next_known = o.gene_index[dummy.contig].lower_bound(dummy.end);
		for (size_t k = 1; ; ++k) {
			if (k == unmapped.size() || dummy.end + 10000 < unmapped[k].second ||
			    (next_known != o.gene_index[dummy.contig].end() && next_known->first <= unmapped[k].second) || unmapped[k].first != dummy.contig) {
				o.genes.push_back(dummy);
				if (k == unmapped.size()) break;
				dummy.contig = unmapped[k].first; dummy.start = unmapped[k].second;
				next_known = o.gene_index[dummy.contig].lower_bound(unmapped[k].second);
			}
			dummy.end = unmapped[k].second;
		}
	}
	// index including the dummy genes (source/arriba.cpp:262-264)
	std::vector<int> contig_of(o.genes.size());
	for (size_t g = 0; g < o.genes.size(); ++g) contig_of[g] = o.genes[g].contig;
	std::vector<bucket_map> index;
	build_index(o.genes, o.genes.size(), o.gene_index.size(), contig_of, index);
	for (size_t i = 0; i < o.reads.size(); ++i) {
		std::vector<Alignment>& a = o.reads[i].a;
		if (a.size() == 3) {
			if (a[0].genes.empty() || a[1].genes.empty()) { int p = breakpoint_inner(a[1]); a[1].genes = features_at(index, a[1].contig, p, p); a[0].genes = a[1].genes; }
			if (a[2].genes.empty()) { int p = breakpoint_outer(a[2]); a[2].genes = features_at(index, a[2].contig, p, p); }
		} else {
			for (size_t s = 0; s < a.size(); ++s) if (a[s].genes.empty()) { int p = breakpoint_outer(a[s]); a[s].genes = features_at(index, a[s].contig, p, p); }
		}
	}
	for (size_t i = 0; i < o.reads.size(); ++i) {
		std::vector<Alignment>& a = o.reads[i].a;
		for (size_t s = 0; s < a.size(); ++s) {
			if (a[s].genes.size() > 1 && o.genes[a[s].genes[0]].dummy) {
				int p = breakpoint_inner(a[s]);
				int chosen = a[0].genes[0];
				for (size_t g = 0; g < a[s].genes.size(); ++g) if (o.genes[a[s].genes[g]].start <= p && o.genes[a[s].genes[g]].end >= p) chosen = a[s].genes[g];
				a[s].genes.assign(1, chosen);
			}
		}
		if (a.size() == 3 && a[0].genes[0] != a[1].genes[0] && o.genes[a[0].genes[0]].dummy && o.genes[a[1].genes[0]].dummy) {
			int p = breakpoint_inner(a[1]);
			int chosen = a[0].genes[0];
			for (int s = 0; s < 2; ++s) for (size_t g = 0; g < a[s].genes.size(); ++g) if (o.genes[a[s].genes[g]].start <= p && o.genes[a[s].genes[g]].end >= p) chosen = a[s].genes[g];
			a[0].genes.assign(1, chosen); a[1].genes.assign(1, chosen);
		}
	}
}

// ---- read-level filters (source/arriba.cpp:327-409) ----------------------------------------------------------------

bool interesting(const oracle_state& o, int contig) { return o.contig_bits[contig] & AGPU_CBIT_INTERESTING; }
bool viral(const oracle_state& o, int contig) { return o.contig_bits[contig] & AGPU_CBIT_VIRAL; }

void boundaries(const oracle_state& o, const gene_list& genes, int& start, int& end) { // source/annotation.cpp:558-567
	start = end = -1;
	for (size_t g = 0; g < genes.size(); ++g) {
		if (start == -1 || start > o.genes[genes[g]].start) start = o.genes[genes[g]].start;
		if (end == -1 || end < o.genes[genes[g]].end) end = o.genes[genes[g]].end;
	}
}

bool within_aligned_segment(const Alignment& a, int breakpoint) { // source/filter_hairpin.cpp:7-27
	int position = a.start;
	for (size_t c = 0; c < a.cigar.size(); ++c) {
		uint32_t op = a.cigar[c] & 15; int length = a.cigar[c] >> 4;
		if (op == 3 || op == 2) position += length;
		else if (op == 0 || op == 8 || op == 7) { if (breakpoint >= position && breakpoint <= position + length) return true; position += length; }
	}
	return false;
}

char complement(char c) { switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; default: return c; } }
std::string reverse_complement(const std::string& s) { std::string r(s.rbegin(), s.rend()); for (size_t i = 0; i < r.size(); ++i) r[i] = complement(r[i]); return r; }

double binomial_coefficient(const unsigned int k, const unsigned int n) { double r = 1; for (unsigned int i = n - k + 1; i <= n; ++i) r *= i; for (unsigned int i = 1; i <= k; ++i) r /= i; return r; }
float binomial_distribution(const unsigned int k, const unsigned int n, const float p) { return binomial_coefficient(k, n) * pow(p, k) * pow(1 - p, n - k); }

// reference: count_mismatches + test_mismatch_probability, source/filter_mismatches.cpp:12-99
bool too_many_mismatches(const oracle_state& o, const Alignment& a, const std::string& sequence, unsigned long genome_size, bool multimapper) {
	unsigned int mismatches = 0, length = 0;
	int reference_position = a.start; unsigned int read_position = 0;
	const std::string& contig = o.genome[a.contig];
	for (size_t c = 0; c < a.cigar.size(); ++c) {
		uint32_t op = a.cigar[c] & 15; unsigned int n = a.cigar[c] >> 4;
		if (op == 4 || op == 5) { read_position += n; if (!((c == 0 && !a.forward) || (c == a.cigar.size() - 1 && a.forward))) mismatches++; }
		else if (op == 2) { mismatches++; reference_position += n; }
		else if (op == 3) reference_position += n;
		else if (op == 1) { mismatches++; read_position += n; }
		else if (op == 0 || op == 7 || op == 8)
			for (unsigned int k = 0; k < n; ++k, ++reference_position, ++read_position)
				if (sequence[read_position] != 'N') {
					char base = (reference_position >= 0 && (size_t) reference_position < contig.size()) ? contig[reference_position] : '\0';
					if (sequence[read_position] != base) mismatches++;
					length++;
				}
	}
	if (multimapper) mismatches += 2;
	const float probability = 0.01, cutoff = o.params.mismatch_pvalue_cutoff;
	if (binomial_distribution(mismatches, length, probability) < cutoff) return true;
	if (mismatches > 0) {
		long double permutations = pow(4, length - mismatches);
		if (genome_size >= permutations) return true;
		return (1 - pow(1 - genome_size / permutations, binomial_coefficient(mismatches, length))) > 0.01;
	}
	return false;
}

unsigned int kmer_code(const std::string& s, size_t position, int k) { // source/filter_mismappers.cpp:33-45
	unsigned int code = 0;
	for (int b = 0; b < k; ++b) { code <<= 2; char c = s[position + b]; code += (c == 'T') ? 0 : (c == 'G') ? 1 : (c == 'C') ? 2 : 3; }
	return code;
}

// reference: source/filter_low_entropy.cpp:33-101
bool low_entropy(const oracle_state& o, const Read& r) {
	const unsigned int k = 3; const float content = o.params.max_kmer_content;
	for (int mate = 0; mate <= 1; ++mate) {
		const std::string& sequence = r.a[mate].sequence;
		if (sequence.length() < k) continue;
		unsigned int s1 = ((r.a[mate].cigar.front() & 15) == 4) ? r.a[mate].cigar.front() >> 4 : 0, e1 = sequence.length();
		if ((r.a[mate].cigar.back() & 15) == 4) e1 -= r.a[mate].cigar.back() >> 4;
		unsigned int s2 = s1, e2 = e1;
		if (r.a.size() == 3 && mate == 1) {
			s2 = ((r.a[2].cigar.front() & 15) == 4) ? r.a[2].cigar.front() >> 4 : 0; e2 = sequence.length();
			if ((r.a[2].cigar.back() & 15) == 4) e2 -= r.a[2].cigar.back() >> 4;
			if (r.a[2].forward != r.a[1].forward) { s2 = sequence.length() - s2; e2 = sequence.length() - e2; std::swap(s2, e2); }
		}
		std::vector<unsigned int> all(64), in1(64), in2(64);
		std::vector<size_t> previous(64);
		unsigned int max_all = sequence.length() * content / k + 0.5, max1 = (e1 - s1) * content / k + 0.5, max2 = (e2 - s2) * content / k + 0.5;
		for (size_t p = 0; p < sequence.length() - k; ++p) {
			unsigned int code = kmer_code(sequence, p, k);
			if (previous[code] > p) continue;
			previous[code] = p + k;
			++all[code];
			if (p + 1 >= s1 && p < e1) ++in1[code];
			if (p + 1 >= s2 && p < e2) ++in2[code];
			if (all[code] >= max_all || in1[code] >= max1 || in2[code] >= max2) return true;
		}
	}
	return false;
}

void read_filters(oracle_state& o, const uint8_t* top_verdict, const uint8_t* low_verdict) {
	const uint8_t* on = o.params.filter_enabled;
	std::vector<Read>& reads = o.reads;
	o.remaining.assign(AGPU_FILTER_COUNT, 0);
	auto count_remaining = [&](int filter) { uint64_t n = 0; for (size_t i = 0; i < reads.size(); ++i) if (reads[i].filter == F_NONE) ++n; o.remaining[filter] = n; };
	// duplicates, source/filter_duplicates.cpp:8-55
	if (on[F_DUPLICATES]) {
		std::set<std::tuple<int,int,int,int> > seen;
		for (size_t i = 0; i < reads.size(); ++i) {
			Read& r = reads[i];
			if (r.filter != F_NONE) continue;
			if (o.params.external_duplicate_marking) { if (r.duplicate) r.filter = F_DUPLICATES; continue; }
			const Alignment& m1 = r.a[0]; const Alignment& m2 = r.a[r.a.size() == 2 ? 1 : 2];
			int p1 = m1.forward ? m1.start - m1.preclip() : m1.end + m1.postclip(), p2 = m2.forward ? m2.start - m2.preclip() : m2.end + m2.postclip();
			int c1 = m1.contig, c2 = m2.contig;
			if (p1 > p2) { std::swap(p1, p2); std::swap(c1, c2); }
			if (!seen.insert(std::make_tuple(c1, c2, p1, p2)).second) r.filter = F_DUPLICATES;
		}
		count_remaining(F_DUPLICATES);
	}
	for (size_t i = 0; i < reads.size(); ++i) { // contig based filters, source/filter_uninteresting_contigs.cpp, filter_viral_contigs.cpp
		Read& r = reads[i];
		if (on[F_UNINTERESTING] && r.filter == F_NONE) for (size_t s = 0; s < r.a.size(); ++s) if (!interesting(o, r.a[s].contig)) { r.filter = F_UNINTERESTING; break; }
	}
	count_remaining(F_UNINTERESTING);
	for (size_t i = 0; i < reads.size(); ++i) {
		Read& r = reads[i];
		if (on[F_VIRAL] && r.filter == F_NONE) { bool all = true; for (size_t s = 0; s < r.a.size(); ++s) if (!viral(o, r.a[s].contig)) all = false; if (all) r.filter = F_VIRAL; }
	}
	count_remaining(F_VIRAL);
	for (size_t i = 0; i < reads.size(); ++i) {
		Read& r = reads[i];
		if (on[F_TOP_VIRAL] && top_verdict && r.filter == F_NONE) for (size_t s = 0; s < r.a.size(); ++s) if (viral(o, r.a[s].contig) && top_verdict[r.a[s].contig]) { r.filter = F_TOP_VIRAL; break; }
	}
	count_remaining(F_TOP_VIRAL);
	for (size_t i = 0; i < reads.size(); ++i) {
		Read& r = reads[i];
		if (on[F_LOW_COV_VIRAL] && low_verdict && r.filter == F_NONE) for (size_t s = 0; s < r.a.size(); ++s) if (viral(o, r.a[s].contig) && low_verdict[r.a[s].contig]) { r.filter = F_LOW_COV_VIRAL; break; }
	}
	count_remaining(F_LOW_COV_VIRAL);

	unsigned long genome_size = 0;
	for (size_t c = 0; c < o.genome.size(); ++c) if (interesting(o, c)) genome_size += o.genome[c].size();
	const int min_distance = o.params.min_read_through_distance;
	const unsigned int H = o.params.homopolymer_length;

	if (on[F_READ_THROUGH]) for (size_t i = 0; i < reads.size(); ++i) { // source/filter_proximal_read_through.cpp:8-47
		Read& r = reads[i]; if (r.filter != F_NONE) continue;
		const Alignment* f; const Alignment* v;
		if (r.a.size() == 2) { f = r.a[0].forward ? &r.a[0] : &r.a[1]; v = r.a[0].forward ? &r.a[1] : &r.a[0]; }
		else { f = r.a[1].forward ? &r.a[2] : &r.a[1]; v = r.a[1].forward ? &r.a[1] : &r.a[2]; }
		bool candidate = f->contig == v->contig && f->end < v->start && ((r.a.size() == 2 && f->forward != v->forward) || (r.a.size() == 3 && f->forward == v->forward));
		if (!candidate) continue;
		int fs, fe, vs, ve; boundaries(o, f->genes, fs, fe); boundaries(o, v->genes, vs, ve);
		if (f->end >= vs - min_distance || v->start <= fe + min_distance) r.filter = F_READ_THROUGH;
	}
	count_remaining(F_READ_THROUGH);
	if (on[F_INCONSISTENTLY_CLIPPED]) for (size_t i = 0; i < reads.size(); ++i) { // source/filter_inconsistently_clipped.cpp:6-25
		Read& r = reads[i]; if (r.filter != F_NONE || r.a.size() != 3) continue;
		if ((r.a[0].forward && r.a[0].end > r.a[1].end + 3) || (!r.a[0].forward && r.a[0].start < r.a[1].start - 3)) r.filter = F_INCONSISTENTLY_CLIPPED;
	}
	count_remaining(F_INCONSISTENTLY_CLIPPED);
	if (on[F_HOMOPOLYMER]) for (size_t i = 0; i < reads.size(); ++i) { // source/filter_homopolymer.cpp:16-62
		Read& r = reads[i]; if (r.filter != F_NONE || r.a.size() != 3) continue;
		const Alignment& split = r.a[1];
		std::string window;
		if (split.forward) {
			if ((unsigned) split.preclip() >= H) window += split.sequence.substr(split.preclip() - H, H) + " ";
			if (split.sequence.length() - split.preclip() >= H) window += split.sequence.substr(split.preclip(), H) + " ";
		} else {
			if ((unsigned) split.postclip() >= H) window += split.sequence.substr(split.sequence.length() - split.postclip(), H) + " ";
			if (split.sequence.length() - split.postclip() >= H) window += split.sequence.substr(split.sequence.length() - split.postclip() - H, H) + " ";
		}
		unsigned int run = 1;
		for (size_t c = 1; c < window.length(); ++c) {
			if (window[c - 1] == window[c]) {
				if (++run == H) {
					bool spliced = false;
					for (size_t g = 0; g < split.genes.size(); ++g) if (breakpoint_spliced(o, split.genes[g], split.forward, breakpoint_inner(split))) spliced = true;
					if (!spliced) { r.filter = F_HOMOPOLYMER; break; }
				}
			} else run = 1;
		}
	}
	count_remaining(F_HOMOPOLYMER);
	if (on[F_SMALL_INSERT]) for (size_t i = 0; i < reads.size(); ++i) { // source/filter_small_insert_size.cpp:7-30
		Read& r = reads[i]; if (r.filter != F_NONE || r.a.size() != 2) continue;
		if (r.a[0].forward != r.a[1].forward && r.a[0].contig == r.a[1].contig && (std::abs(r.a[0].start - r.a[1].start) <= 5 || std::abs(r.a[0].end - r.a[1].end) <= 5)) r.filter = F_SMALL_INSERT;
	}
	count_remaining(F_SMALL_INSERT);
	if (on[F_LONG_GAP]) for (size_t i = 0; i < reads.size(); ++i) { // source/filter_long_gap.cpp:6-89
		Read& r = reads[i]; if (r.filter != F_NONE) continue;
		int deletion = 0;
		if (r.a.size() == 3 && r.a[1].contig == r.a[2].contig) {
			if (!r.a[1].forward && !r.a[2].forward) deletion = r.a[2].start - r.a[1].end;
			else if (r.a[1].forward && r.a[2].forward) deletion = r.a[1].start - r.a[2].end;
		}
		for (size_t s = 0; s < r.a.size() && r.filter == F_NONE; ++s) {
			const std::vector<uint32_t>& cigar = r.a[s].cigar;
			for (size_t c = 1; c + 1 < cigar.size(); ++c) {
				if ((cigar[c] & 15) != 3 || !((int) (cigar[c] >> 4) >= 700000 || (deletion >= 700000 && deletion <= 1500000))) continue;
				unsigned int left = 0, right = 0;
				for (int j = (int) c - 1; j >= 0; --j) { uint32_t op = cigar[j] & 15; if (op == 0 || op == 8 || op == 7) left += cigar[j] >> 4; else if (op != 2 && op != 1 && op != 6) break; }
				for (size_t j = c + 1; j < cigar.size(); ++j) { uint32_t op = cigar[j] & 15; if (op == 0 || op == 8 || op == 7) right += cigar[j] >> 4; else if (op != 2 && op != 1 && op != 6) break; }
				if (left <= 15 && right <= 15) { r.filter = F_LONG_GAP; break; }
			}
		}
	}
	count_remaining(F_LONG_GAP);
	if (on[F_SAME_GENE]) for (size_t i = 0; i < reads.size(); ++i) { // source/filter_same_gene.cpp:8-46
		Read& r = reads[i]; if (r.filter != F_NONE) continue;
		gene_list common = (r.a.size() == 2) ? set_intersection(r.a[0].genes, r.a[1].genes) : set_intersection(r.a[1].genes, r.a[2].genes);
		if (common.empty()) continue;
		if (r.a.size() == 2) {
			if ((r.a[0].forward && !r.a[1].forward && r.a[0].start <= r.a[1].end) || (!r.a[0].forward && r.a[1].forward && r.a[0].end >= r.a[1].start)) r.filter = F_SAME_GENE;
		} else if ((r.a[1].forward && r.a[2].forward && r.a[1].start >= r.a[2].end) || (!r.a[1].forward && !r.a[2].forward && r.a[1].end <= r.a[2].start)) r.filter = F_SAME_GENE;
	}
	count_remaining(F_SAME_GENE);
	if (on[F_HAIRPIN]) for (size_t i = 0; i < reads.size(); ++i) { // source/filter_hairpin.cpp:29-80
		Read& r = reads[i]; if (r.filter != F_NONE) continue;
		if (r.a.size() == 2) {
			if (set_intersection(r.a[0].genes, r.a[1].genes).empty() && r.a[0].contig != r.a[1].contig) continue;
			if (within_aligned_segment(r.a[1], breakpoint_outer(r.a[0])) || within_aligned_segment(r.a[0], breakpoint_outer(r.a[1]))) r.filter = F_HAIRPIN;
		} else {
			if (set_intersection(r.a[1].genes, r.a[2].genes).empty() && r.a[1].contig != r.a[2].contig) continue;
			int split = breakpoint_inner(r.a[1]), supplementary = breakpoint_outer(r.a[2]);
			if (within_aligned_segment(r.a[2], split) || within_aligned_segment(r.a[1], supplementary) || within_aligned_segment(r.a[0], supplementary)) r.filter = F_HAIRPIN;
		}
	}
	count_remaining(F_HAIRPIN);
	if (on[F_MISMATCHES]) for (size_t i = 0; i < reads.size(); ++i) { // source/filter_mismatches.cpp:101-135
		Read& r = reads[i]; if (r.filter != F_NONE) continue;
		const Alignment& other = r.a[r.a.size() == 2 ? 1 : 2];
		bool discard = !viral(o, r.a[0].contig) && too_many_mismatches(o, r.a[0], r.a[0].sequence, genome_size, r.multimapper && !viral(o, other.contig));
		if (!discard && !viral(o, other.contig)) {
			std::string sequence = (r.a.size() == 2) ? r.a[1].sequence : ((r.a[2].forward == r.a[1].forward) ? r.a[1].sequence : reverse_complement(r.a[1].sequence));
			discard = too_many_mismatches(o, other, sequence, genome_size, r.multimapper && !viral(o, r.a[0].contig));
		}
		if (discard) r.filter = F_MISMATCHES;
	}
	count_remaining(F_MISMATCHES);
	if (on[F_LOW_ENTROPY]) for (size_t i = 0; i < reads.size(); ++i) { // source/filter_low_entropy.cpp:11-31
		Read& r = reads[i];
		const int max_itd = o.params.max_itd_length;
		bool itd = r.a.size() == 3 && r.a[1].forward == r.a[2].forward && r.a[1].contig == r.a[2].contig &&
		           ((r.a[1].forward && r.a[1].start < r.a[2].end && r.a[1].start + max_itd >= r.a[2].end) || (!r.a[1].forward && r.a[1].end > r.a[2].start && r.a[1].end <= r.a[2].start + max_itd));
		if ((!itd || r.filter == F_DUPLICATES) && r.filter != F_NONE) continue;
		if (low_entropy(o, r)) r.filter = F_LOW_ENTROPY;
	}
	count_remaining(F_LOW_ENTROPY);
}

// ---- find_fusions (source/fusions.cpp:15-473) ------------------------------------------------------------------------

typedef std::tuple<int,int,int,int,int,int,bool,bool> candidate_key;

bool is_intragenic(const oracle_state& o, const Candidate& f) { // source/common.hpp:275-279
	return f.gene1 == f.gene2 || (f.breakpoint1 >= o.genes[f.gene2].start - 10000 && f.breakpoint1 <= o.genes[f.gene2].end + 10000 && f.breakpoint2 >= o.genes[f.gene1].start - 10000 && f.breakpoint2 <= o.genes[f.gene1].end + 10000);
}

void update_anchor(int& anchor, int value, bool upstream) { // source/fusions.cpp:276-285
	if (!upstream && (value < anchor || anchor == 0)) anchor = value;
	else if (upstream && (value > anchor || anchor == 0)) anchor = value;
}

void find_fusions(oracle_state& o, int max_mate_gap) {
	const unsigned int T = o.params.subsampling_threshold;
	std::map<candidate_key, size_t> lookup;
	std::vector<Candidate>& candidates = o.candidates;
	candidates.clear();
	o.discordant_swapped.assign(o.reads.size(), 0);
	typedef std::tuple<int,int,bool,bool> pair_key;
	std::map<pair_key, std::vector<std::tuple<int,int,int> > > discordant_by_gene_pair;
	for (size_t i = 0; i < o.reads.size(); ++i) {
		const Read& r = o.reads[i];
		bool split = r.a.size() == 3;
		const Alignment& first = split ? r.a[1] : r.a[0];
		const Alignment& second = split ? r.a[2] : r.a[1];
		int contig1 = first.contig, contig2 = second.contig;
		int breakpoint1 = split ? breakpoint_inner(first) : breakpoint_outer(first), breakpoint2 = breakpoint_outer(second);
		bool upstream1 = split ? first.forward : !first.forward, upstream2 = !second.forward;
		gene_list genes1 = first.genes, genes2 = second.genes;
		bool exonic1 = first.exonic, exonic2 = second.exonic;
		int anchor1 = r.a[0].forward ? r.a[0].start : r.a[0].end, anchor2 = second.forward ? second.start : second.end;
		bool swapped = false;
		if (contig1 > contig2 || (contig1 == contig2 && breakpoint1 > breakpoint2)) {
			std::swap(contig1, contig2); std::swap(breakpoint1, breakpoint2); std::swap(genes1, genes2); std::swap(upstream1, upstream2); std::swap(exonic1, exonic2); std::swap(anchor1, anchor2);
			swapped = true;
		}
		for (size_t g1 = 0; g1 < genes1.size(); ++g1) for (size_t g2 = 0; g2 < genes2.size(); ++g2) {
			candidate_key key(genes1[g1], genes2[g2], contig1, contig2, breakpoint1, breakpoint2, upstream1, upstream2);
			std::pair<std::map<candidate_key, size_t>::iterator, bool> inserted = lookup.insert(std::make_pair(key, candidates.size()));
			if (inserted.second) {
				Candidate fresh; fresh.gene1 = genes1[g1]; fresh.gene2 = genes2[g2]; fresh.contig1 = contig1; fresh.contig2 = contig2; fresh.breakpoint1 = breakpoint1; fresh.breakpoint2 = breakpoint2;
				fresh.upstream1 = upstream1; fresh.upstream2 = upstream2;
				candidates.push_back(fresh);
			}
			Candidate& f = candidates[inserted.first->second];
			f.exonic1 = f.exonic1 || exonic1; f.exonic2 = f.exonic2 || exonic2;
			if (inserted.second || r.filter == F_NONE || f.filter == F_DUPLICATES) f.filter = r.filter;
			if (split) {
				bool subsampled = (!swapped && (unsigned) f.split_reads1 >= T) || (swapped && (unsigned) f.split_reads2 >= T) ||
				                  (r.filter != F_NONE && !swapped && f.list1.size() >= T) || (r.filter != F_NONE && swapped && f.list2.size() >= T);
				if (subsampled) continue;
				update_anchor(f.anchor1, anchor1, f.upstream1); update_anchor(f.anchor2, anchor2, f.upstream2);
				if (swapped) { f.list2.push_back(i); if (r.filter == F_NONE) f.split_reads2++; } else { f.list1.push_back(i); if (r.filter == F_NONE) f.split_reads1++; }
			} else {
				update_anchor(f.anchor1, anchor1, f.upstream1); update_anchor(f.anchor2, anchor2, f.upstream2);
				discordant_by_gene_pair[pair_key(genes1[g1], genes2[g2], upstream1, upstream2)].push_back(std::make_tuple(breakpoint1, breakpoint2, (int) i));
			}
		}
	}
	for (size_t c = 0; c < candidates.size(); ++c) { // source/fusions.cpp:367-437
		Candidate& f = candidates[c];
		if (f.filter != F_NONE) continue;
		std::map<pair_key, std::vector<std::tuple<int,int,int> > >::iterator bucket = discordant_by_gene_pair.find(pair_key(f.gene1, f.gene2, f.upstream1, f.upstream2));
		if (bucket == discordant_by_gene_pair.end()) continue;
		int overlap = (f.list1.size() + f.list2.size() > 0) ? 2 : max_mate_gap;
		int limit1 = f.upstream1 ? f.breakpoint1 - overlap : f.breakpoint1 + overlap, limit2 = f.upstream2 ? f.breakpoint2 - overlap : f.breakpoint2 + overlap;
		for (size_t k = 0; k < bucket->second.size(); ++k) {
			int mate1 = std::get<0>(bucket->second[k]), mate2 = std::get<1>(bucket->second[k]), read = std::get<2>(bucket->second[k]);
			if (!(f.upstream1 ? mate1 >= limit1 : mate1 <= limit1) || !(f.upstream2 ? mate2 >= limit2 : mate2 <= limit2)) continue;
			bool far = !is_intragenic(o, f) && !(mate1 >= o.genes[f.gene2].start && mate1 <= o.genes[f.gene2].end) && !(mate2 >= o.genes[f.gene1].start && mate2 <= o.genes[f.gene1].end);
			if (!(far || (std::abs(f.breakpoint1 - mate1) <= max_mate_gap && std::abs(f.breakpoint2 - mate2) <= max_mate_gap))) continue;
			Read& r = o.reads[read];
			if (r.filter != F_NONE && f.discordant.size() >= T) continue;
			if ((unsigned) f.discordant_mates >= T) break;
			f.discordant.push_back(read);
			if (r.filter == F_NONE) f.discordant_mates++;
			Alignment& m1 = r.a[0]; Alignment& m2 = r.a[1];
			if (m1.contig > m2.contig || (m1.contig == m2.contig && breakpoint_outer(m1) > breakpoint_outer(m2))) { std::swap(m1, m2); o.discordant_swapped[read] ^= 1; }
			update_anchor(f.anchor1, f.upstream1 ? m1.end : m1.start, f.upstream1);
			update_anchor(f.anchor2, f.upstream2 ? m2.end : m2.start, f.upstream2);
		}
	}
	for (size_t c = 0; c < candidates.size(); ++c) { // strands, splice sites, transcript start: source/fusions.cpp:15-200, 443-470
		Candidate& f = candidates[c];
		unsigned int forward = 0, reverse = 0;
		for (size_t k = 0; k < f.list1.size(); ++k) { const Alignment& a = o.reads[f.list1[k]].a[1]; if (!a.predicted_ambiguous) { if (a.predicted_forward) ++forward; else ++reverse; } }
		for (size_t k = 0; k < f.list2.size(); ++k) { const Alignment& a = o.reads[f.list2[k]].a[2]; if (!a.predicted_ambiguous) { if (a.predicted_forward) ++forward; else ++reverse; } }
		for (size_t k = 0; k < f.discordant.size(); ++k) {
			const Read& r = o.reads[f.discordant[k]];
			if (r.a[0].predicted_ambiguous || r.filter == F_HAIRPIN) continue;
			const Alignment* m1 = &r.a[0]; const Alignment* m2 = &r.a[1];
			if (m1->contig != f.contig1 || (m1->forward != !f.upstream1)) std::swap(m1, m2);
			else if (m1->forward == m2->forward) {
				int e1 = f.upstream1 ? m1->start : m1->end, e2 = f.upstream1 ? m2->start : m2->end;
				unsigned int d1 = std::abs(f.breakpoint1 - e1) + std::abs(f.breakpoint2 - e2), d2 = std::abs(f.breakpoint2 - e1) + std::abs(f.breakpoint1 - e2);
				if (d1 == d2) continue;
				if (d2 < d1) std::swap(m1, m2);
			}
			if (m1->predicted_forward) ++forward; else ++reverse;
		}
		if (forward == reverse) f.strands_ambiguous = true;
		else { f.strands_ambiguous = false; f.strand1 = forward > reverse; f.strand2 = flip_if(f.strand1, f.upstream1 == f.upstream2); }
		const Gene& g1 = o.genes[f.gene1]; const Gene& g2 = o.genes[f.gene2];
		if (f.list1.size() + f.list2.size() == 0 || f.strands_ambiguous) f.spliced1 = f.spliced2 = false;
		else {
			f.spliced1 = f.exonic1 && g1.forward == f.strand1 && breakpoint_spliced(o, f.gene1, f.upstream1, f.breakpoint1);
			f.spliced2 = f.exonic2 && g2.forward == f.strand2 && breakpoint_spliced(o, f.gene2, f.upstream2, f.breakpoint2);
		}
		bool read_through = f.contig1 == f.contig2 && f.breakpoint2 - f.breakpoint1 < 400000 && !f.upstream1 && f.upstream2;
		f.start_ambiguous = false;
		auto sense1 = [&]() { return (g1.forward && !f.upstream1) || (!g1.forward && f.upstream1); };
		auto sense2 = [&]() { return (g2.forward && !f.upstream2) || (!g2.forward && f.upstream2); };
		if (f.spliced1 || (!f.strands_ambiguous && !g1.dummy && f.strand1 == g1.forward)) f.start_gene1 = sense1();
		else if (f.spliced2 || (!f.strands_ambiguous && !g2.dummy && f.strand2 == g2.forward)) f.start_gene1 = !sense2();
		else if (!f.strands_ambiguous) {
			bool out1 = (f.strand1 && !f.upstream1) || (!f.strand1 && f.upstream1), in2 = (!f.strand2 && !f.upstream2) || (f.strand2 && f.upstream2);
			bool out2 = (f.strand2 && !f.upstream2) || (!f.strand2 && f.upstream2), in1 = (!f.strand1 && !f.upstream1) || (f.strand1 && f.upstream1);
			if (out1 && in2) f.start_gene1 = true; else if (out2 && in1) f.start_gene1 = false; else f.start_ambiguous = true;
		} else if (!f.exonic1 && !f.exonic2) f.start_ambiguous = true;
		else if (!f.exonic1 && f.exonic2) {
			if (sense2()) f.start_gene1 = false;
			else if (f.split_reads1 + f.split_reads2 == 0 && read_through) f.start_gene1 = true;
			else f.start_ambiguous = true;
		} else if (!f.exonic2 && f.exonic1) {
			if (sense1()) f.start_gene1 = true;
			else if (f.split_reads1 + f.split_reads2 == 0 && read_through) f.start_gene1 = true;
			else f.start_ambiguous = true;
		} else {
			if ((!g1.dummy && g1.forward && !f.upstream1) || (!g1.forward && f.upstream1)) f.start_gene1 = true;
			else if ((!g2.dummy && g2.forward && !f.upstream2) || (!g2.forward && f.upstream2)) f.start_gene1 = false;
			else f.start_ambiguous = true;
		}
		if (f.start_ambiguous) f.start_gene1 = true;
		if (!f.start_ambiguous && f.strands_ambiguous) {
			f.strands_ambiguous = false;
			if (f.start_gene1) { f.strand1 = g1.forward; f.strand2 = flip_if(f.strand1, f.upstream1 == f.upstream2); }
			else { f.strand2 = g2.forward; f.strand1 = flip_if(f.strand2, f.upstream1 == f.upstream2); }
		}
	}
}

}

extern "C" {

oracle_state* oracle_create(const agpu_params* params, const agpu_annotation_view* annotation, const agpu_genome_view* genome, const agpu_batch_view* batch) {
	oracle_state* o = new oracle_state();
	o->params = *params;
	o->real_genes = annotation->n_genes;
	for (uint32_t g = 0; g < annotation->n_genes; ++g) {
		Gene gene; gene.contig = annotation->gene_contig[g]; gene.start = annotation->gene_start[g]; gene.end = annotation->gene_end[g];
		gene.forward = annotation->gene_bits[g] & AGPU_GBIT_STRAND; gene.dummy = annotation->gene_bits[g] & AGPU_GBIT_DUMMY;
		o->genes.push_back(gene);
	}
	std::vector<int> exon_contig(annotation->n_exons), gene_contig(annotation->n_genes);
	for (uint32_t e = 0; e < annotation->n_exons; ++e) {
		Exon exon; exon.start = annotation->exon_start[e]; exon.end = annotation->exon_end[e]; exon.gene = annotation->exon_gene[e]; exon.previous = annotation->exon_previous[e]; exon.next = annotation->exon_next[e];
		exon.cds_start = annotation->exon_cds_start[e]; exon.cds_end = annotation->exon_cds_end[e];
		o->exons.push_back(exon);
		exon_contig[e] = o->genes[exon.gene].contig;
	}
	for (uint32_t g = 0; g < annotation->n_genes; ++g) gene_contig[g] = o->genes[g].contig;
	size_t n_contigs = std::max<size_t>(genome->n_contigs, annotation->gene_index.n_contigs);
	build_index(o->exons, o->exons.size(), n_contigs, exon_contig, o->exon_index);
	build_index(o->genes, o->genes.size(), n_contigs, gene_contig, o->gene_index);
	o->genome.resize(genome->n_contigs); o->contig_bits.assign(genome->contig_bits, genome->contig_bits + genome->n_contigs);
	for (uint32_t c = 0; c < genome->n_contigs; ++c) o->genome[c].assign((const char*) genome->bases + genome->contig_offset[c], genome->contig_offset[c + 1] - genome->contig_offset[c]);
	static const char codes[] = "=ACMGRSVTWYHKDBN";
	o->reads.resize(batch->n);
	for (uint64_t i = 0; i < batch->n; ++i) {
		Read& r = o->reads[i];
		r.single_end = batch->fbits[i] & AGPU_FBIT_SINGLE_END; r.duplicate = batch->fbits[i] & AGPU_FBIT_DUPLICATE; r.multimapper = false; r.filter = F_NONE; r.group = batch->group[i];
		r.a.resize(batch->n_aln[i]);
		for (size_t s = 0; s < r.a.size(); ++s) {
			Alignment& a = r.a[s];
			uint8_t bits = batch->abits[s][i];
			a.supplementary = bits & AGPU_ABIT_SUPPLEMENTARY; a.first_in_pair = bits & AGPU_ABIT_FIRST_IN_PAIR; a.forward = bits & AGPU_ABIT_STRAND; a.exonic = false; a.predicted_forward = false; a.predicted_ambiguous = true;
			a.contig = batch->contig[s][i]; a.start = batch->start[s][i]; a.end = batch->end[s][i];
			a.cigar.assign(batch->cigar_pool + batch->cigar_offset[s][i], batch->cigar_pool + batch->cigar_offset[s][i] + batch->cigar_count[s][i]);
			if (s < 2) {
				const uint8_t* packed = batch->seq_pool + (size_t) batch->seq_offset[s][i] * 4;
				a.sequence.resize(batch->seq_length[s][i]);
				for (size_t b = 0; b < a.sequence.size(); ++b) a.sequence[b] = codes[(packed[b >> 1] >> ((~b & 1) << 2)) & 15];
			}
		}
	}
	return o;
}
void oracle_destroy(oracle_state* o) { delete o; }

// mark_multimappers (source/read_chimeric_alignments.cpp:792-802) + annotation (source/arriba.cpp:160-325)
uint64_t oracle_annotate(oracle_state* o, int strandedness) {
	o->params.strandedness = strandedness;
	uint64_t marked = 0;
	for (size_t i = 0; i + 1 < o->reads.size(); ++i)
		if (o->reads[i].group == o->reads[i + 1].group) { o->reads[i].multimapper = o->reads[i + 1].multimapper = true; ++marked; }
	for (size_t i = 0; i < o->reads.size(); ++i) annotate_read(*o, o->reads[i]);
	fallback_and_dummy_genes(*o);
	return marked;
}
void oracle_read_filters(oracle_state* o, const uint8_t* top_verdict, const uint8_t* low_verdict, uint64_t* remaining) {
	read_filters(*o, top_verdict, low_verdict);
	if (remaining) memcpy(remaining, o->remaining.data(), AGPU_FILTER_COUNT * sizeof(uint64_t));
}
uint64_t oracle_find_fusions(oracle_state* o, int max_mate_gap) { find_fusions(*o, max_mate_gap); return o->candidates.size(); }

uint32_t oracle_gene_count(oracle_state* o) { return o->genes.size(); }
void oracle_get_gene(oracle_state* o, uint32_t g, int32_t* fields /* contig, start, end, forward, dummy */) {
	const Gene& gene = o->genes[g]; fields[0] = gene.contig; fields[1] = gene.start; fields[2] = gene.end; fields[3] = gene.forward; fields[4] = gene.dummy;
}
void oracle_get_filters(oracle_state* o, uint8_t* filters) { for (size_t i = 0; i < o->reads.size(); ++i) filters[i] = o->reads[i].filter; }
void oracle_get_alignment_bits(oracle_state* o, int slot, uint8_t* bits) {
	for (size_t i = 0; i < o->reads.size(); ++i) {
		bits[i] = 0;
		if ((size_t) slot >= o->reads[i].a.size()) continue;
		// find_fusions swaps the mates of attached discordant fragments in place; report the original slot order
		int s = (o->reads[i].a.size() == 2 && !o->discordant_swapped.empty() && o->discordant_swapped[i]) ? 1 - slot : slot;
		const Alignment& a = o->reads[i].a[s];
		bits[i] = (a.forward ? AGPU_ABIT_STRAND : 0) | (a.first_in_pair ? AGPU_ABIT_FIRST_IN_PAIR : 0) | (a.supplementary ? AGPU_ABIT_SUPPLEMENTARY : 0) | (a.exonic ? AGPU_ABIT_EXONIC : 0) |
		          (a.predicted_ambiguous ? AGPU_ABIT_PREDICTED_STRAND_AMBIGUOUS : (a.predicted_forward ? AGPU_ABIT_PREDICTED_STRAND : 0));
	}
}
void oracle_get_fragment_bits(oracle_state* o, uint8_t* bits) {
	for (size_t i = 0; i < o->reads.size(); ++i) bits[i] = (o->reads[i].single_end ? AGPU_FBIT_SINGLE_END : 0) | (o->reads[i].multimapper ? AGPU_FBIT_MULTIMAPPER : 0) | (o->reads[i].duplicate ? AGPU_FBIT_DUPLICATE : 0);
}
uint64_t oracle_get_gene_sets(oracle_state* o, int slot, uint8_t* count, uint32_t* genes, uint64_t capacity) {
	uint64_t total = 0;
	for (size_t i = 0; i < o->reads.size(); ++i) {
		size_t n = 0;
		if ((size_t) slot < o->reads[i].a.size()) {
			int s = (o->reads[i].a.size() == 2 && !o->discordant_swapped.empty() && o->discordant_swapped[i]) ? 1 - slot : slot;
			const gene_list& list = o->reads[i].a[s].genes;
			n = list.size();
			if (genes) for (size_t g = 0; g < n && total + g < capacity; ++g) genes[total + g] = list[g];
		}
		if (count) count[i] = n;
		total += n;
	}
	return total;
}
// candidate c: 12 scalar fields + flags in the AGPU_CFLAG_* encoding + list sizes
void oracle_get_candidate(oracle_state* o, uint64_t c, int64_t* fields) {
	const Candidate& f = o->candidates[c];
	uint32_t flags = (f.upstream1 ? AGPU_CFLAG_UPSTREAM1 : 0) | (f.upstream2 ? AGPU_CFLAG_UPSTREAM2 : 0) | (f.exonic1 ? AGPU_CFLAG_EXONIC1 : 0) | (f.exonic2 ? AGPU_CFLAG_EXONIC2 : 0) |
	                 (f.spliced1 ? AGPU_CFLAG_SPLICED1 : 0) | (f.spliced2 ? AGPU_CFLAG_SPLICED2 : 0) | (f.strand1 ? AGPU_CFLAG_PREDICTED_STRAND1 : 0) | (f.strand2 ? AGPU_CFLAG_PREDICTED_STRAND2 : 0) |
	                 (f.strands_ambiguous ? AGPU_CFLAG_PREDICTED_STRANDS_AMBIGUOUS : 0) | (f.start_gene1 ? AGPU_CFLAG_TRANSCRIPT_START_GENE1 : 0) | (f.start_ambiguous ? AGPU_CFLAG_TRANSCRIPT_START_AMBIGUOUS : 0);
	int64_t values[16] = { f.gene1, f.gene2, ((int64_t) f.contig1 << 16) | f.contig2, f.breakpoint1, f.breakpoint2, flags, f.filter, f.split_reads1, f.split_reads2, f.discordant_mates, f.anchor1, f.anchor2,
	                       (int64_t) f.list1.size(), (int64_t) f.list2.size(), (int64_t) f.discordant.size(), 0 };
	memcpy(fields, values, sizeof(values));
}
void oracle_get_candidate_lists(oracle_state* o, uint64_t c, uint32_t* reads) {
	const Candidate& f = o->candidates[c];
	size_t at = 0;
	for (size_t k = 0; k < f.list1.size(); ++k) reads[at++] = f.list1[k];
	for (size_t k = 0; k < f.list2.size(); ++k) reads[at++] = f.list2[k];
	for (size_t k = 0; k < f.discordant.size(); ++k) reads[at++] = f.discordant[k];
}
void oracle_get_discordant_swapped(oracle_state* o, uint8_t* swapped) { memcpy(swapped, o->discordant_swapped.data(), o->reads.size()); }

}
