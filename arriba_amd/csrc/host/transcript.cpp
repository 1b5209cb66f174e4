// arriba_amd/csrc/host/transcript.cpp -- the columns of fusions.tsv that come from the supporting reads and the transcript annotation: the fusion
// transcript assembled from the pileup of the reads at the two breakpoints (reference: source/output_fusions.cpp:25-466), the annotated
// transcripts that fit it best (:711-818), the peptide and its reading frame (source/annotate_protein_domains.cpp:163-446).  Host code over
// the few candidates that pass all filters; every quirk of the reference's string handling is kept because the columns are compared byte
// for byte.  fill_gaps_in_fusion_transcript (-I, :820-1041) completes the assembled sequence with the reference genome along the chosen transcripts.
//
// One place of the reference cannot be reproduced in general: get_transcripts walks an unordered_map keyed by heap pointers
// (source/output_fusions.cpp:792-804), so its choice between transcripts with mixed coding status depends on addresses (it differs between
// two runs of the reference itself with address-space randomisation).  Here the transcripts are walked in ascending id.
#include "transcript.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

namespace arriba {

thread_local std::string* transcript_warnings = NULL; // set by the output writer around the formatting of one row
// (ARRIBA_WRITER_PROFILE: where the threads spend the time of fusion_transcript_sequence -- pile-ups, columns of the pile-ups, the consensus over them, the rest)
std::atomic<long long> transcript_profile_ns[4];
namespace {
struct ProfileLap {
	const bool on; std::chrono::steady_clock::time_point mark;
	ProfileLap(): on(getenv("ARRIBA_WRITER_PROFILE") != NULL && getenv("ARRIBA_WRITER_PROFILE")[0] == '2'), mark(on ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point()) {}
	void lap(int part) { if (!on) return; const std::chrono::steady_clock::time_point now = std::chrono::steady_clock::now(); transcript_profile_ns[part] += std::chrono::duration_cast<std::chrono::nanoseconds>(now - mark).count(); mark = now; }
};
}



namespace {

enum { CIGAR_MATCH = 0, CIGAR_INSERTION = 1, CIGAR_DELETION = 2, CIGAR_SKIP = 3, CIGAR_SOFT_CLIP = 4, CIGAR_HARD_CLIP = 5, CIGAR_EQUAL = 7, CIGAR_DIFF = 8 };

char complement_of(char base) {
	switch (base) {
		case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c';
		case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
		case '[': return ']'; case ']': return '[';
		default: return base;
	}
}
std::string reverse_complement(const std::string& dna) {
	std::string result(dna.size(), 'N');
	for (size_t i = 0; i < dna.size(); ++i) result[dna.size() - 1 - i] = complement_of(dna[i]);
	return result;
}
bool is_intron_allele(const std::string& allele) { return allele == "_" || allele == ">" || allele == "<"; }
bool is_intron_character(char allele) { return allele == '_' || allele == '>' || allele == '<'; }
bool is_lower_case_base(char c) { return c == 'a' || c == 't' || c == 'c' || c == 'g'; }

// The pileup next to a breakpoint: position -> allele -> reads.  Alleles: a base, "-" (deleted), an insertion + the base behind it, and ">" "_" "<" for the
// start / inside / end of an intron.  The reference keeps it in nested std::maps keyed by std::string (source/output_fusions.cpp:25-107) and pays two tree look-ups
// per base of every supporting read and one tree node per position of every intron -- ~0.5 ms per fusion, seconds for the best-supported ones.  Here the
// single-character alleles (all but the insertions) are counted in pages of 256 positions x 20 alleles (the last page is remembered: consecutive bases fall on
// consecutive positions); what the consensus walks over is the same: the positions in ascending order, at each the alleles in the order of their strings.
const char* const SINGLE_ALLELES = "-<=>ABCDGHKMNRSTVWY_"; // in ASCII order == the order of std::map<std::string, ...>
const int N_SINGLE_ALLELES = 20;
struct Allele { // `single` = the allele if it is one character (all but insertions), 0 otherwise with `multi` = its string (owned by the pileup): the consensus compares characters, not strings
	const std::string* multi; unsigned count; char single;
	Allele(const std::string& t, unsigned c): multi(t.size() == 1 ? NULL : &t), count(c), single(t.size() == 1 ? t[0] : 0) {}
	Allele(char c, unsigned n): multi(NULL), count(n), single(c) {}
	std::string text() const { return multi != NULL ? *multi : std::string(1, single); }
};
struct Column { position_t position; size_t first, n; }; // alleles[first .. first + n)
// The pages of the pileups of a thread are kept and handed out again: a row of the output file piles up ~10 pages of 20 KB and gives them back, all writer threads at once;
// through malloc / free the arenas of the threads grow and shrink by that much per row, and every shrink is an madvise that interrupts all cores (measured on the 256-thread
// host of the GPU box: 4 ms per row on 128 threads where one thread needs 0.2 ms -- profiles/r03c_mismapper_second_pass.txt, [writer] lines).
const size_t PILEUP_PAGE_WORDS = 256 * 20 + 8 + 256; // the counters of 256 positions x 20 alleles; a bit per position that holds any; per position a bit per allele that was added (columns() visits only those)
struct PilePagePool {
	std::vector<unsigned*> free_pages;
	~PilePagePool() { for (size_t k = 0; k < free_pages.size(); ++k) delete[] free_pages[k]; }
	unsigned* take() { unsigned* page; if (free_pages.empty()) page = new unsigned[PILEUP_PAGE_WORDS]; else { page = free_pages.back(); free_pages.pop_back(); } memset(page, 0, PILEUP_PAGE_WORDS * sizeof(unsigned)); return page; }
	void give(unsigned* page) { free_pages.push_back(page); }
};
thread_local PilePagePool pile_page_pool;
class DensePileup {
public:
	DensePileup(): last_page_(0), last_(NULL) { for (int c = 0; c < 256; ++c) slot_of_char_[c] = -1; for (int k = 0; k < N_SINGLE_ALLELES; ++k) slot_of_char_[(unsigned char) SINGLE_ALLELES[k]] = k; }
	~DensePileup() { for (std::map<position_t, unsigned*>::iterator page = pages_.begin(); page != pages_.end(); ++page) pile_page_pool.give(page->second); }
	DensePileup(const DensePileup&) = delete; DensePileup& operator=(const DensePileup&) = delete;
	int slot_of(char allele) const { return slot_of_char_[(unsigned char) allele]; }
	void add(position_t position, int slot, unsigned count = 1) {
		const position_t page = position >> 8;
		if (last_ == NULL || page != last_page_) { unsigned*& counts = pages_[page]; if (counts == NULL) counts = pile_page_pool.take(); last_ = counts; last_page_ = page; }
		last_[(size_t) (position & 255) * N_SINGLE_ALLELES + slot] += count;
		last_[256 * N_SINGLE_ALLELES + ((position & 255) >> 5)] |= 1u << (position & 31);
		last_[256 * N_SINGLE_ALLELES + 8 + (position & 255)] |= 1u << slot;
	}
	// single-character alleles at consecutive positions from `position` on, slot_at(k) = the slot of the k-th: the pages are looked up once per page, not once per base
	template <class SlotAt> void add_run(position_t position, int n, SlotAt slot_at) {
		for (int k = 0; k < n; ) {
			const position_t page = position >> 8;
			if (last_ == NULL || page != last_page_) { unsigned*& counts = pages_[page]; if (counts == NULL) counts = pile_page_pool.take(); last_ = counts; last_page_ = page; }
			const int first = (int) (position & 255), in_page = std::min(n - k, 256 - first);
			unsigned* row = last_ + (size_t) first * N_SINGLE_ALLELES; unsigned* masks = last_ + 256 * N_SINGLE_ALLELES + 8 + first;
			for (int i = 0; i < in_page; ++i, row += N_SINGLE_ALLELES) { const int slot = slot_at(k + i); row[slot]++; masks[i] |= 1u << slot; }
			for (int at = first; at < first + in_page; ) { // the bits of the touched positions, a word at a time
				const int word = at >> 5, upto = std::min(first + in_page, (word + 1) << 5);
				const unsigned bits = (upto - at == 32) ? 0xFFFFFFFFu : (((1u << (upto - at)) - 1u) << (at & 31));
				last_[256 * N_SINGLE_ALLELES + word] |= bits;
				at = upto;
			}
			k += in_page; position += in_page;
		}
	}
	void add(position_t position, const std::string& allele) { // any allele (insertions; what std::string::substr gives at the end of a sequence)
		if (allele.size() == 1 && slot_of(allele[0]) >= 0) add(position, slot_of(allele[0])); else other_[position][allele]++;
	}
	void add_inside_intron(position_t from, position_t to, unsigned count) { if (from <= to && count > 0) { Interval interval = { from, to, count }; inside_introns_.push_back(interval); } } // "_" at every position of [from, to]
	// The columns in ascending order of their positions, the alleles of a column in the order of their strings.  The insides of introns are kept as intervals: a
	// stretch of positions that hold nothing but the same number of "_" is a stretch of identical columns, of which the consensus only ever singles out the first or
	// the last one (the trimming by coverage) and acts on the first one (the intron opens) -- so only these two are made, however long the intron is.
	void columns(std::vector<Column>& columns, std::vector<Allele>& alleles) const {
		columns.clear(); alleles.clear();
		// the positions that hold counted alleles, ascending
		static thread_local std::vector<std::pair<position_t, const unsigned*> > counted; // (kept between the rows of a thread, like the pages)
		counted.clear();
		for (std::map<position_t, unsigned*>::const_iterator page = pages_.begin(); page != pages_.end(); ++page)
			for (int word = 0; word < 8; ++word)
				for (unsigned touched = page->second[256 * N_SINGLE_ALLELES + word]; touched != 0; touched &= touched - 1) {
					const int at = 32 * word + __builtin_ctz(touched);
					counted.push_back(std::make_pair(page->first * 256 + at, page->second + (size_t) at * N_SINGLE_ALLELES));
				}
		static thread_local std::vector<std::pair<position_t, long long> > events; // the number of reads inside an intron changes by .second at position .first
		events.clear();
		for (size_t k = 0; k < inside_introns_.size(); ++k) { events.push_back(std::make_pair(inside_introns_[k].from, (long long) inside_introns_[k].count)); events.push_back(std::make_pair(inside_introns_[k].to + 1, -(long long) inside_introns_[k].count)); }
		std::sort(events.begin(), events.end());
		std::map<position_t, std::map<std::string, unsigned> >::const_iterator other = other_.begin();
		size_t c = 0, e = 0;
		long long inside = 0;          // reads inside an intron at the positions from `unemitted` on
		position_t unemitted = 0; bool have_unemitted = false;
		auto emit_inside_only = [&](position_t from, position_t to) {
			if (inside <= 0 || from > to) return;
			for (int k = 0; k < (to > from ? 2 : 1); ++k) { Column column = { k == 0 ? from : to, alleles.size(), 1 }; alleles.push_back(Allele('_', (unsigned) inside)); columns.push_back(column); }
		};
		while (c < counted.size() || e < events.size() || other != other_.end()) {
			position_t at = 0; bool have = false;
			if (c < counted.size()) { at = counted[c].first; have = true; }
			if (e < events.size() && (!have || events[e].first < at)) { at = events[e].first; have = true; }
			if (other != other_.end() && (!have || other->first < at)) { at = other->first; have = true; }
			if (have_unemitted) emit_inside_only(unemitted, at - 1);
			unemitted = at; have_unemitted = true;
			while (e < events.size() && events[e].first == at) inside += events[e++].second;
			const bool counted_here = c < counted.size() && counted[c].first == at, other_here = other != other_.end() && other->first == at;
			if (!counted_here && !other_here) continue; // (only the number of reads inside introns changed here: the stretch from `at` on is made when its end is known)
			Column column = { at, alleles.size(), 0 };
			if (counted_here) { // the alleles added at this position, in the order of their characters (a bit per allele behind the counters of the page)
				const unsigned* counts = counted[c].second;
				const unsigned* page = counts - (size_t) (at & 255) * N_SINGLE_ALLELES;
				for (unsigned slots = page[256 * N_SINGLE_ALLELES + 8 + (at & 255)]; slots != 0; slots &= slots - 1) { const int slot = __builtin_ctz(slots); if (counts[slot] > 0) alleles.push_back(Allele(SINGLE_ALLELES[slot], counts[slot])); }
				++c;
			}
			if (inside > 0) { // ("_" may have been counted here as an ordinary allele, too: one allele, the sum)
				bool merged = false;
				for (size_t a = column.first; a < alleles.size() && !merged; ++a) if (alleles[a].single == '_') { alleles[a].count += (unsigned) inside; merged = true; }
				if (!merged) alleles.push_back(Allele('_', (unsigned) inside)); // ('_' is the last of the single alleles in the order of their strings: still sorted)
			}
			if (other_here) { // (rare: insertions, and what substr gives at the end of a sequence) these go where their strings belong
				for (std::map<std::string, unsigned>::const_iterator a = other->second.begin(); a != other->second.end(); ++a) alleles.push_back(Allele(a->first, a->second));
				++other;
				std::sort(alleles.begin() + column.first, alleles.end(), [](const Allele& x, const Allele& y) { return x.text() < y.text(); });
			}
			column.n = alleles.size() - column.first;
			columns.push_back(column);
			unemitted = at + 1;
		}
	}
private:
	std::map<position_t, unsigned*> pages_;
	std::map<position_t, std::map<std::string, unsigned> > other_;
	struct Interval { position_t from, to; unsigned count; };
	std::vector<Interval> inside_introns_;
	position_t last_page_; unsigned* last_;
	int slot_of_char_[256];
};

struct Reads { // the fragments of the sample and their final filters
	const Batch& batch; const uint8_t* filter;
	bool strand(unsigned slot, uint32_t read) const { return batch.abits[slot][read] & ABIT_STRAND; }
	unsigned clipping(unsigned slot, uint32_t read, bool front) const { // preclipping / postclipping (source/common.hpp:205-206)
		const uint32_t count = batch.cigar_count[slot][read];
		const uint32_t element = batch.cigar_pool[batch.cigar_offset[slot][read] + (front ? 0 : count - 1)];
		return ((element & 15) == CIGAR_SOFT_CLIP || (element & 15) == CIGAR_HARD_CLIP) ? element >> 4 : 0;
	}
};

// reference: pileup_chimeric_alignments (:25-107): the alignments in slot `mate` of the fragments list[0 .. n)
void add_to_pileup(const Reads& reads, const uint32_t* list, uint32_t n, unsigned mate, bool reverse, bool upstream, position_t breakpoint, DensePileup& pileup) {
	const Batch& b = reads.batch;
	const int deletion_slot = pileup.slot_of('-');
	int slot_of_code[16];
	for (int code = 0; code < 16; ++code) slot_of_code[code] = pileup.slot_of("=ACMGRSVTWYHKDBN"[code]);
	std::map<std::pair<position_t, position_t>, unsigned> introns;
	for (uint32_t k = 0; k < n; ++k) {
		const uint32_t read = list[k];
		if (reads.filter[read] == FILTER_duplicates) continue;
		const position_t start = b.start[mate][read], end = b.end[mate][read];
		const bool forward = reads.strand(mate, read), is_split_read = b.n_aln[read] == 3;
		if (!is_split_read && // discordant mates: only those close to the breakpoint, distant ones may belong to other isoforms
		    !((!upstream && forward && end <= breakpoint + 2 && end >= breakpoint - 200) || (upstream && !forward && start >= breakpoint - 2 && start <= breakpoint + 200))) continue;
		if (is_split_read && (mate == SPLIT_READ || mate == SUPPLEMENTARY) && start != breakpoint && end != breakpoint) continue; // alternative alignments with shifted breakpoints
		// the bases of the read stay packed (4 bits each, "=ACMGRSVTWYHKDBN"): the pile-up takes the slot of a code from a table; the characters are only made for the rare
		// alleles that are strings (insertions, what substr gives at the end of the sequence)
		const unsigned sequence_slot = mate == SUPPLEMENTARY ? SPLIT_READ : mate;
		const uint8_t* packed = &b.seq_pool[0] + (size_t) b.seq_offset[sequence_slot][read] * 4;
		const size_t sequence_length = b.seq_length[sequence_slot][read];
		static const uint8_t complement_code[16] = { 0, 8, 4, 3, 2, 5, 6, 7, 1, 9, 10, 11, 12, 13, 14, 15 }; // A <-> T, C <-> G, the ambiguity codes as they are (complement_of)
		auto slot_at = [&](size_t i) -> int {
			const size_t j = reverse ? sequence_length - 1 - i : i;
			const unsigned code = (packed[j >> 1] >> ((~j & 1) << 2)) & 15;
			return slot_of_code[reverse ? complement_code[code] : code];
		};
		static thread_local std::string sequence; // (kept between the reads of a thread; filled when a string is needed)
		bool have_sequence = false;
		auto need_sequence = [&]() -> const std::string& {
			if (!have_sequence) {
				b.sequence_into(sequence_slot, read, sequence);
				if (reverse) { std::reverse(sequence.begin(), sequence.end()); for (size_t c = 0; c < sequence.size(); ++c) sequence[c] = complement_of(sequence[c]); }
				have_sequence = true;
			}
			return sequence;
		};
		position_t read_offset = 0, reference_offset = start;
		int borrowed = 0; // an insertion takes one base of the next element along
		const uint32_t* cigar = &b.cigar_pool[b.cigar_offset[mate][read]];
		const unsigned elements = b.cigar_count[mate][read];
		for (unsigned e = 0; e < elements; ++e) {
			const unsigned operation = cigar[e] & 15; const int length = (int) (cigar[e] >> 4);
			bool consume_bases = false;
			switch (operation) {
				case CIGAR_INSERTION:
					pileup.add(reference_offset, need_sequence().substr(read_offset, length + 1));
					read_offset += length + 1; ++reference_offset; borrowed = 1;
					break;
				case CIGAR_SKIP: {
					const position_t intron_start = reference_offset;
					reference_offset += length - borrowed;
					introns[std::make_pair(intron_start, reference_offset - 1)]++;
					borrowed = 0;
					break;
				}
				case CIGAR_DELETION:
					for (int base = 0; base < length - borrowed; ++base, ++reference_offset) pileup.add(reference_offset, deletion_slot);
					borrowed = 0;
					break;
				case CIGAR_HARD_CLIP:
					if (mate == SUPPLEMENTARY) read_offset += length;
					break;
				case CIGAR_SOFT_CLIP:
					// the clipped segment of a split read at its breakpoint is piled up, too: non-template bases show there
					if (is_split_read && mate == SPLIT_READ && ((e == 0 && forward) || (e == elements - 1 && !forward))) {
						if (e == 0 && forward) reference_offset -= length;
						consume_bases = true;
					} else
						read_offset += length - borrowed;
					break;
				case CIGAR_MATCH: case CIGAR_EQUAL: case CIGAR_DIFF:
					consume_bases = true;
					break;
			}
			if (consume_bases) {
				const int bases = length - borrowed;
				const int inside = read_offset < 0 ? 0 : (int) std::min<long long>(bases, (long long) sequence_length - read_offset); // bases that lie inside the sequence
				if (inside > 0) { const position_t from = read_offset; pileup.add_run(reference_offset, inside, [&](int k) { return slot_at((size_t) (from + k)); }); }
				// (std::string::substr semantics of the reference: a position at the end of the sequence gives the empty allele, beyond it is an error)
				for (int base = inside > 0 ? inside : 0; base < bases; ++base) pileup.add(reference_offset + base, need_sequence().substr(read_offset + base, 1));
				read_offset += bases; reference_offset += bases;
				borrowed = 0;
			}
		}
	}
	for (std::map<std::pair<position_t, position_t>, unsigned>::const_iterator intron = introns.begin(); intron != introns.end(); ++intron) {
		pileup.add(intron->first.first, pileup.slot_of('>'), intron->second);
		pileup.add(intron->first.second, pileup.slot_of('<'), intron->second);
		pileup.add_inside_intron(intron->first.first + 1, intron->first.second - 1, intron->second);
	}
}

unsigned reads_at(const std::vector<Allele>& alleles, const Column& column) {
	unsigned total = 0;
	for (size_t a = column.first; a < column.first + column.n; ++a) total += alleles[a].count;
	return total;
}

// reference: get_sequence_from_pileup (:109-240): the consensus next to one breakpoint; `clipped` receives what lies beyond the breakpoint
void consensus_of_pileup(const DensePileup& pileup, position_t breakpoint, bool upstream, contig_t contig, const Assembly& assembly, std::string& sequence, std::vector<position_t>& positions, std::string& clipped) {
	static thread_local std::vector<Column> columns; static thread_local std::vector<Allele> alleles; // (cleared and filled by columns(): their memory stays with the thread)
	{ ProfileLap profile; pileup.columns(columns, alleles); profile.lap(1); }
	unsigned peak = 0;
	for (size_t at = 0; at < columns.size(); ++at) peak = std::max(peak, reads_at(alleles, columns[at]));
	// thin coverage far from the breakpoint probably belongs to other isoforms
	const float low_coverage_fraction = 0.10;
	size_t first = 0, last = columns.size();
	for (size_t at = 0; at < columns.size(); ++at) {
		const unsigned coverage = reads_at(alleles, columns[at]);
		if (!upstream) { if (coverage < peak * low_coverage_fraction) first = at; else break; }
		else if (coverage > peak * low_coverage_fraction) last = at;
	}
	if (last != columns.size()) ++last;
	bool intron_open = false, intron_closed = true;
	for (size_t at = first; at < last; ++at) {
		const Column& column = columns[at];
		if (at != first && columns[at - 1].position < column.position - 1 && !intron_open) { sequence += "..."; positions.resize(positions.size() + 3, -1); } // not covered
		char reference = 'N';
		if (assembly.has(contig) && (unsigned) column.position < assembly.sequence[contig].size()) reference = assembly.sequence[contig][column.position];
		// the most frequent allele; ties go to the reference base, and to introns before anything else
		const Allele* best = NULL;
		unsigned coverage = 0;
		for (size_t a = column.first; a < column.first + column.n; ++a) {
			const Allele* allele = &alleles[a];
			const char c = allele->single;
			if (best == NULL || allele->count > best->count ||
			    (allele->count == best->count && ((c == reference && !is_intron_character(best->single)) || (c == '<' && best->single != '_' && best->single != '>') || c == '_' || c == '>')))
				best = allele;
			if (!is_intron_character(c)) coverage += allele->count;
		}
		// trusted: >= 75 % of the reads, an intron at least as frequent as the coverage, or the reference base
		const bool trusted = (is_intron_character(best->single) && best->count >= coverage) || best->count >= 0.75 * coverage || best->single == reference;
		const char single = trusted ? best->single : '?';
		if (single == '_') {
			if (!intron_open) { sequence += "...___"; positions.resize(positions.size() + 6, -1); intron_open = true; intron_closed = false; } // inside an intron whose start was not seen
		} else if (single == '>') {
			if (!intron_open) { sequence += "___"; positions.resize(positions.size() + 3, -1); intron_open = true; intron_closed = false; }
		} else if (single == '<') {
			if (!intron_open) { sequence += "...___"; positions.resize(positions.size() + 6, -1); }
			intron_open = true; intron_closed = true;
		} else {
			if (!intron_closed) { sequence += "..."; positions.resize(positions.size() + 3, -1); } // the end of the intron was not seen
			intron_open = false; intron_closed = true;
			const bool upstream_of_breakpoint = (upstream && column.position < breakpoint) || (!upstream && column.position > breakpoint);
			if (single != 0) { // one base (or '?'): a mismatch in lower case
				const char base = (single != reference && reference != 'N') ? (char) tolower(single) : single;
				if (upstream_of_breakpoint) clipped += base;
				else { sequence += base; positions.push_back(column.position); }
			} else { // an insertion (or the empty allele of a read that ends here), as a string: in lower case, the inserted bases in brackets, then the base behind them
				std::string allele = best->text();
				if (allele.size() > 1 || (allele != std::string(1, reference) && reference != 'N'))
					for (size_t i = 0; i < allele.size(); ++i) allele[i] = (char) tolower(allele[i]);
				if (allele.size() > 1) {
					allele = "[" + allele.substr(0, allele.size() - 1) + "]" + allele[allele.size() - 1];
					positions.resize(positions.size() + allele.size() - 1, -1);
					if (toupper(allele[allele.size() - 1]) == reference) allele[allele.size() - 1] = (char) toupper(allele[allele.size() - 1]);
				}
				if (upstream_of_breakpoint) clipped += allele;
				else { sequence += allele; positions.push_back(column.position); }
			}
		}
	}
}

// mismatched (lower-case) bases right at the breakpoint are usually non-template bases: they are split off with a pipe (:340-391)
bool split_off_non_template_bases(bool upstream, std::string& sequence, std::vector<position_t>& positions) {
	if (upstream) {
		int base = 0;
		while (base < (int) sequence.size() && is_lower_case_base(sequence[base])) ++base;
		if (base > 0 && base < (int) sequence.size()) {
			sequence = sequence.substr(0, base) + "|" + sequence.substr(base);
			std::fill(positions.begin(), positions.begin() + base, -1);
			positions.insert(positions.begin() + base, -1);
			return true;
		}
	} else {
		int base = (int) sequence.size() - 1;
		while (base >= 0 && is_lower_case_base(sequence[base])) --base;
		if (base + 1 < (int) sequence.size() && base >= 0) {
			sequence = sequence.substr(0, base + 1) + "|" + sequence.substr(base + 1);
			std::fill(positions.begin() + base + 1, positions.end(), -1);
			positions.insert(positions.begin() + base + 1, -1);
			return true;
		}
	}
	return false;
}

}

// reference: get_fusion_transcript_sequence (:242-466)
void fusion_transcript_sequence(const TranscriptInput& in, const FusionEvent& f, std::string& sequence, std::vector<position_t>& positions) {
	ProfileLap profile;
	sequence.clear(); positions.clear();
	if (f.strands_ambiguous || f.transcript_start_ambiguous) { sequence = "."; positions.push_back(-1); return; } // the strands are unknown
	const Reads reads = { in.batch, in.read_filter };
	const uint32_t* list1 = f.split_read1_list; const uint32_t* list2 = f.split_read2_list; const uint32_t* mates = f.discordant_mate_list;
	DensePileup pileup1, pileup2;
	add_to_pileup(reads, list1, f.n_split_reads1, SPLIT_READ, false, f.upstream1, f.breakpoint1, pileup1);
	add_to_pileup(reads, list1, f.n_split_reads1, MATE1, false, f.upstream1, f.breakpoint1, pileup1);
	add_to_pileup(reads, list1, f.n_split_reads1, SUPPLEMENTARY, f.upstream1 == f.upstream2, f.upstream2, f.breakpoint2, pileup2);
	add_to_pileup(reads, list2, f.n_split_reads2, SPLIT_READ, false, f.upstream2, f.breakpoint2, pileup2);
	add_to_pileup(reads, list2, f.n_split_reads2, MATE1, false, f.upstream2, f.breakpoint2, pileup2);
	add_to_pileup(reads, list2, f.n_split_reads2, SUPPLEMENTARY, f.upstream1 == f.upstream2, f.upstream1, f.breakpoint1, pileup1);
	add_to_pileup(reads, mates, f.n_discordant_mates, MATE1, false, f.upstream1, f.breakpoint1, pileup1);
	add_to_pileup(reads, mates, f.n_discordant_mates, MATE2, false, f.upstream1, f.breakpoint1, pileup1);
	add_to_pileup(reads, mates, f.n_discordant_mates, MATE1, false, f.upstream2, f.breakpoint2, pileup2);
	add_to_pileup(reads, mates, f.n_discordant_mates, MATE2, false, f.upstream2, f.breakpoint2, pileup2);
	profile.lap(0);

	// non-template bases between the genes: the clipped bases of split read and supplementary alignment add up to more than the read; the most frequent count wins, the first one in list order on a tie
	unsigned non_template_bases = 0;
	std::map<unsigned, unsigned> reads_by_count;
	for (uint32_t k = 0; k < f.n_split_reads1 + f.n_split_reads2; ++k) {
		const uint32_t read = k < f.n_split_reads1 ? list1[k] : list2[k - f.n_split_reads1];
		const unsigned clipped_split_read = reads.clipping(SPLIT_READ, read, reads.strand(SPLIT_READ, read)), clipped_supplementary = reads.clipping(SUPPLEMENTARY, read, !reads.strand(SUPPLEMENTARY, read));
		const unsigned read_length = in.batch.seq_length[SPLIT_READ][read];
		if (clipped_split_read + clipped_supplementary >= read_length) {
			const unsigned unmapped = clipped_split_read + clipped_supplementary - read_length;
			if (++reads_by_count[unmapped] > reads_by_count[non_template_bases]) non_template_bases = unmapped;
		}
	}

	std::string sequence1, sequence2, clipped1, clipped2;
	std::vector<position_t> positions1, positions2;
	consensus_of_pileup(pileup1, f.breakpoint1, f.upstream1, f.contig_of_gene1, in.assembly, sequence1, positions1, clipped1);
	consensus_of_pileup(pileup2, f.breakpoint2, f.upstream2, f.contig_of_gene2, in.assembly, sequence2, positions2, clipped2);
	profile.lap(2);

	if (f.n_split_reads1 + f.n_split_reads2 == 0) { // without split reads the exact breakpoints are unknown
		if (!f.upstream1) { sequence1 += "..."; positions1.resize(positions1.size() + 3, -1); } else { sequence1 = "..." + sequence1; positions1.insert(positions1.begin(), 3, -1); }
		if (!f.upstream2) { sequence2 += "..."; positions2.resize(positions2.size() + 3, -1); } else { sequence2 = "..." + sequence2; positions2.insert(positions2.begin(), 3, -1); }
	}

	if (non_template_bases > 0) {
		std::string* clipped = clipped1.size() >= non_template_bases ? &clipped1 : clipped2.size() >= non_template_bases ? &clipped2 : NULL;
		if (clipped != NULL) {
			const bool first = clipped == &clipped1;
			std::string& target = first ? sequence1 : sequence2; std::vector<position_t>& target_positions = first ? positions1 : positions2;
			for (size_t i = 0; i < clipped->size(); ++i) (*clipped)[i] = (char) tolower((*clipped)[i]);
			if (first ? f.upstream1 : f.upstream2) { target = clipped->substr(clipped->size() - non_template_bases) + target; target_positions.insert(target_positions.begin(), non_template_bases, -1); }
			else { target += clipped->substr(0, non_template_bases); target_positions.resize(target_positions.size() + non_template_bases, -1); }
		}
	}

	const bool split1 = split_off_non_template_bases(f.upstream1, sequence1, positions1), split2 = split_off_non_template_bases(f.upstream2, sequence2, positions2);

	// 5' part, junction, 3' part, in the direction of transcription
	std::string* head = f.transcript_start_gene1 ? &sequence1 : &sequence2; std::string* tail = f.transcript_start_gene1 ? &sequence2 : &sequence1;
	std::vector<position_t>* head_positions = f.transcript_start_gene1 ? &positions1 : &positions2; std::vector<position_t>* tail_positions = f.transcript_start_gene1 ? &positions2 : &positions1;
	const bool head_forward = f.transcript_start_gene1 ? f.predicted_strand1 : f.predicted_strand2, tail_upstream = f.transcript_start_gene1 ? f.upstream2 : f.upstream1;
	sequence = head_forward ? *head : reverse_complement(*head);
	if (!head_forward) std::reverse(head_positions->begin(), head_positions->end());
	positions = *head_positions;
	if (!split1 || !split2) { sequence += "|"; positions.push_back(-1); } // otherwise there could be three pipes
	sequence += tail_upstream ? *tail : reverse_complement(*tail);
	if (!tail_upstream) std::reverse(tail_positions->begin(), tail_positions->end());
	positions.insert(positions.end(), tail_positions->begin(), tail_positions->end());

	// "...A...", "...AA..." and the like become "..." (:419-431)
	const size_t max_bases_between_ellipses = 10;
	size_t first_ellipsis = 0, second_ellipsis = std::string::npos;
	while ((first_ellipsis = sequence.find("...", first_ellipsis)) < sequence.size()) {
		if ((second_ellipsis = sequence.find("...", first_ellipsis + 3)) < first_ellipsis + max_bases_between_ellipses + 3 && sequence.find('|', first_ellipsis + 3) > second_ellipsis) { // never across the junction
			sequence.replace(first_ellipsis + 3, second_ellipsis - first_ellipsis, "");
			positions.erase(positions.begin() + first_ellipsis + 3, positions.begin() + second_ellipsis + 3);
		} else first_ellipsis += 3;
	}
	// regions of uncertainty (:433-455): the first pattern of the list that occurs is rewritten, until none occurs
	static const char* const simplifications[][2] = { { "...___|", "|" }, { "|___...", "|" }, { "___|", "...|" }, { "|___", "|..." }, { "______", "___" }, { "___...___", "___" }, { "...___...", "..." }, { "......", "..." } };
	bool rewritten = true;
	while (rewritten) {
		rewritten = false;
		for (size_t s = 0; s < sizeof(simplifications) / sizeof(simplifications[0]) && !rewritten; ++s) {
			const std::string search = simplifications[s][0], replacement = simplifications[s][1];
			const size_t at = sequence.find(search);
			if (at >= sequence.size()) continue;
			sequence.replace(at, search.size(), replacement);
			if (search.size() > replacement.size()) positions.erase(positions.begin() + at, positions.begin() + at + search.size() - replacement.size());
			rewritten = true;
		}
	}
	while (sequence.substr(0, 3) == "..." || sequence.substr(0, 3) == "___") { sequence = sequence.substr(3); positions.erase(positions.begin(), positions.begin() + 3); }
	while (sequence.size() >= 3 && (sequence.substr(sequence.size() - 3) == "..." || sequence.substr(sequence.size() - 3) == "___")) { sequence = sequence.substr(0, sequence.size() - 3); positions.erase(positions.end() - 3, positions.end()); }
	if (sequence == "" || sequence == "|" || sequence == "...|" || sequence == "|..." || sequence == "...|...") { sequence = "."; positions.clear(); positions.push_back(-1); return; } // nothing assembled
	for (size_t i = 0; i < sequence.size(); ++i) if (sequence[i] == 'n' || sequence[i] == 'N') sequence[i] = '?';
	profile.lap(3);
}

// reference: get_transcripts (:720-818): the annotated transcripts of `gene` whose exons fit the transcribed bases of one end (5 or 3) best
void best_fitting_transcripts(const TranscriptInput& in, const std::string& sequence, const std::vector<position_t>& transcribed_bases, int gene, bool gene_is_dummy, contig_t gene_contig, bool gene_forward, bool strand, bool strand_ambiguous, int which_end, std::vector<int>& best) {
	best.clear();
	if (gene_is_dummy) return; // an intergenic region has no exons: nothing scores
	if (strand_ambiguous || strand != gene_forward) return; // anti-sense transcription
	size_t from, to, breakpoint;
	if (which_end == 5) {
		from = 0;
		to = sequence.find('|');
		if (to >= sequence.size()) return;
		while (to > 0 && transcribed_bases[to] == -1) to--; // control characters
		if (transcribed_bases[to] == -1) return;
		breakpoint = to;
	} else {
		from = sequence.find_last_of('|');
		while (from < sequence.size() && transcribed_bases[from] == -1) from++;
		if (from >= sequence.size()) return;
		breakpoint = from;
		to = sequence.size() - 1;
	}
	if (transcribed_bases[from] > transcribed_bases[to]) std::swap(from, to); // from: the lower genomic coordinate
	const size_t low = std::min(from, to), high = std::max(from, to);
	const Annotation& annotation = in.annotation; const FlatIndex& index = in.exon_index;
	if ((size_t) gene_contig >= index.n_contigs()) return;
	std::map<int, unsigned> score, peak_score, transcribed_utr_bases; std::map<int, bool> coding_at_breakpoint;
	size_t position = from;
	const uint32_t contig_begin = index.contig_begin(gene_contig), contig_end = index.contig_end(gene_contig);
	for (uint32_t bucket = index.lower_bound(gene_contig, transcribed_bases[from]); bucket != contig_end && position >= low && position <= high; ++bucket) {
		const position_t bucket_end = index.keys[bucket];
		const uint32_t* members = &index.members[0] + index.member_offset[bucket]; const uint32_t n_members = index.member_offset[bucket + 1] - index.member_offset[bucket];
		position_t last_transcribed_base = transcribed_bases[to];
		while (position >= low && position <= high && transcribed_bases[position] <= bucket_end) { // +1 for every transcribed base inside an exon
			const position_t base = transcribed_bases[position];
			for (uint32_t m = 0; m < n_members; ++m) {
				const ExonRecord& exon = annotation.exons[members[m]];
				if (exon.gene != gene || base < exon.start || base > exon.end) continue;
				const TranscriptRecord& transcript = annotation.transcripts[exon.transcript];
				const bool terminal_first = (int) members[m] == transcript.first_exon, terminal_last = (int) members[m] == transcript.last_exon;
				score[exon.transcript]++;
				last_transcribed_base = base;
				if (terminal_first || terminal_last) transcribed_utr_bases[exon.transcript]++; // UTR annotation is inaccurate: corrected for below
				if (position == breakpoint) {
					if (base >= exon.coding_region_start && base <= exon.coding_region_end) coding_at_breakpoint[exon.transcript] = true;
					if ((abs(base - exon.start) <= 2 && !terminal_first) || (abs(base - exon.end) <= 2 && !terminal_last)) score[exon.transcript] += 10; // the breakpoint is a splice site
				}
			}
			position += (from <= to) ? +1 : -1;
		}
		for (uint32_t m = 0; m < n_members; ++m) {
			const ExonRecord& exon = annotation.exons[members[m]];
			if (exon.gene != gene) continue;
			peak_score[exon.transcript] = std::max(score[exon.transcript], peak_score[exon.transcript]);
			// -1 for every base of the exon that is not transcribed (unsigned arithmetic as in the reference; never below 0)
			const position_t exon_start = bucket != contig_begin ? index.keys[bucket - 1] : exon.start - 1;
			const unsigned exon_length = std::min(bucket_end, transcribed_bases[to]) - std::max(last_transcribed_base + 1, exon_start) + 1;
			score[exon.transcript] -= std::min(exon_length, score[exon.transcript]);
		}
	}
	if (peak_score.empty()) return;
	best.push_back(peak_score.begin()->first);
	for (std::map<int, unsigned>::const_iterator transcript = std::next(peak_score.begin()); transcript != peak_score.end(); ++transcript) {
		const int current = best[0], candidate = transcript->first;
		if (transcript->second == peak_score[current] && coding_at_breakpoint[current] == coding_at_breakpoint[candidate]) best.push_back(candidate);
		else if (transcript->second > peak_score[current] ||
		         (!coding_at_breakpoint[current] && coding_at_breakpoint[candidate] &&
		          (transcript->second == peak_score[current] || (transcribed_utr_bases[candidate] > 0 && transcribed_utr_bases[current] > 0 && transcript->second - transcribed_utr_bases[candidate] >= peak_score[current] - transcribed_utr_bases[current])))) {
			best.clear(); best.push_back(candidate);
		}
	}
	if (peak_score[best[0]] == 0) best.clear();
	std::sort(best.begin(), best.end(), [&](int x, int y) { // by coding length, then length, then id
		const TranscriptRecord& a = annotation.transcripts[x]; const TranscriptRecord& b = annotation.transcripts[y];
		const int length_a = annotation.exons[a.last_exon].end - annotation.exons[a.first_exon].start, length_b = annotation.exons[b.last_exon].end - annotation.exons[b.first_exon].start;
		if (a.coding_length != b.coding_length) return a.coding_length > b.coding_length;
		if (length_a != length_b) return length_a > length_b;
		return a.id < b.id;
	});
	if (best.size() > 1) best.push_back(best[0]); // the best of them first and last: picked whether or not an in-frame combination turns up
}

// reference: fill_gaps_in_fusion_transcript_sequence (:820-1041): with -I the stretches of the chosen transcripts that the reads do not cover are
// taken from the assembly, in parentheses; "^" and "$" mark a sequence that reaches the start / end of the transcript
void fill_gaps_in_fusion_transcript(const TranscriptInput& in, std::string& sequence, std::vector<position_t>& positions, int transcript_5, int transcript_3, bool strand_5, bool strand_3, bool is_internal_tandem_duplication) {
	const Annotation& a = in.annotation;
	bool five_prime_done = false;
	if (transcript_5 != -1 && in.assembly.has(a.exons[a.transcripts[transcript_5].first_exon].contig)) {
		const TranscriptRecord& transcript = a.transcripts[transcript_5];
		const ExonRecord& first_exon = a.exons[transcript.first_exon]; const ExonRecord& last_exon = a.exons[transcript.last_exon];
		const std::string& genome = in.assembly.sequence[first_exon.contig];
		// the gap closest to the junction: the sequence is completed from there
		const size_t breakpoint = sequence.find('|');
		size_t gap = sequence.find_last_of("...", breakpoint);
		bool imprecise_breakpoint = false, fill = true;
		if (gap < sequence.size() && gap + 1 == breakpoint && gap >= 3) { imprecise_breakpoint = true; gap -= 3; } // "...|": a splice site close by may serve as the breakpoint
		else if (gap < sequence.size() && positions[gap + 1] > first_exon.start && positions[gap + 1] < last_exon.end) gap++; // behind the "..."
		else if (gap >= sequence.size() && positions[0] > first_exon.start && positions[0] < last_exon.end) gap = 0;          // no gap, but the transcribed region does not reach the start
		else {
			// no gaps and the transcript is covered: trim to its boundaries
			for (unsigned int i = 0; i < breakpoint; i++)
				if (positions[i] >= first_exon.start && positions[i] <= last_exon.end) {
					if (i > 0) { sequence = sequence.substr(i); positions.erase(positions.begin(), positions.begin() + i); }
					break;
				}
			if ((strand_5 && positions[0] == first_exon.start) || (!strand_5 && positions[0] == last_exon.end)) { sequence = "^" + sequence; positions.insert(positions.begin(), -1); }
			fill = false; five_prime_done = true;
		}
		if (fill) {
			// a position between gap and junction inside an exon of the transcript
			bool overlap_found = false;
			int overlapping_exon = -1;
			for (; gap != breakpoint; gap++) {
				for (overlapping_exon = transcript.first_exon; overlapping_exon != -1; overlapping_exon = a.exons[overlapping_exon].next_exon)
					if (positions[gap] >= a.exons[overlapping_exon].start && positions[gap] <= a.exons[overlapping_exon].end) { overlap_found = true; break; }
				if (overlap_found) break;
			}
			if (imprecise_breakpoint && ((strand_5 && overlapping_exon == transcript.last_exon) || (!strand_5 && overlapping_exon == transcript.first_exon) || is_internal_tandem_duplication)) overlap_found = false;
			if (overlap_found) {
				if (imprecise_breakpoint) { // pretend the breakpoint is the closest exon boundary
					gap = breakpoint - 1;
					positions[gap] = strand_5 ? a.exons[overlapping_exon].end : a.exons[overlapping_exon].start;
					sequence[gap] = strand_5 ? genome[positions[gap]] : complement_of(genome[positions[gap]]);
				}
				std::string from_assembly = "(";
				std::vector<position_t> assembly_positions(1, -1);
				for (int exon = strand_5 ? transcript.first_exon : transcript.last_exon; exon != -1; exon = strand_5 ? a.exons[exon].next_exon : a.exons[exon].previous_exon) {
					const ExonRecord& e = a.exons[exon];
					position_t position;
					for (position = strand_5 ? e.start : e.end; position != positions[gap] && position >= e.start && position <= e.end; position += strand_5 ? +1 : -1) {
						from_assembly += strand_5 ? genome[position] : complement_of(genome[position]);
						assembly_positions.push_back(position);
					}
					if (position == positions[gap]) break;
					from_assembly += "___";
					assembly_positions.resize(assembly_positions.size() + 3, -1);
				}
				if (imprecise_breakpoint) { from_assembly += sequence[gap]; assembly_positions.push_back(positions[gap]); gap++; }
				from_assembly += ")";
				assembly_positions.push_back(-1);
				from_assembly.append(sequence, gap, std::string::npos);
				assembly_positions.insert(assembly_positions.end(), positions.begin() + gap, positions.end());
				sequence = from_assembly; positions = assembly_positions;
				if ((strand_5 && positions[1] == first_exon.start) || (!strand_5 && positions[1] == last_exon.end)) { sequence = "^" + sequence; positions.insert(positions.begin(), -1); }
			}
		}
	}
	(void) five_prime_done;
	if (transcript_3 != -1 && in.assembly.has(a.exons[a.transcripts[transcript_3].first_exon].contig)) {
		const TranscriptRecord& transcript = a.transcripts[transcript_3];
		const ExonRecord& first_exon = a.exons[transcript.first_exon]; const ExonRecord& last_exon = a.exons[transcript.last_exon];
		const std::string& genome = in.assembly.sequence[first_exon.contig];
		const size_t breakpoint = sequence.find_last_of('|');
		size_t gap = sequence.find("...", breakpoint);
		bool imprecise_breakpoint = false;
		if (gap < sequence.size() && gap - 1 == breakpoint && gap + 3 < sequence.size()) { imprecise_breakpoint = true; gap += 3; }
		else if (gap < sequence.size() && positions[gap - 1] > first_exon.start && positions[gap - 1] < last_exon.end) gap--; // the last position in front of "..."
		else if (gap >= sequence.size() && positions[sequence.size() - 1] > first_exon.start && positions[sequence.size() - 1] < last_exon.end) gap = sequence.size() - 1;
		else {
			for (unsigned int i = sequence.size() - 1; i > breakpoint; i--)
				if (positions[i] >= first_exon.start && positions[i] <= last_exon.end) {
					if (i < sequence.size() - 1) { sequence = sequence.substr(0, i + 1); positions.erase(positions.begin() + i + 1, positions.end()); }
					break;
				}
			if ((strand_3 && positions[positions.size() - 1] == last_exon.end) || (!strand_3 && positions[positions.size() - 1] == first_exon.start)) { sequence += "$"; positions.push_back(-1); }
			return;
		}
		bool overlap_found = false;
		int overlapping_exon = -1;
		for (; gap != breakpoint; gap--) {
			for (overlapping_exon = transcript.first_exon; overlapping_exon != -1; overlapping_exon = a.exons[overlapping_exon].next_exon)
				if (positions[gap] >= a.exons[overlapping_exon].start && positions[gap] <= a.exons[overlapping_exon].end) { overlap_found = true; break; }
			if (overlap_found) break;
		}
		if (imprecise_breakpoint && ((strand_3 && overlapping_exon == transcript.last_exon) || (!strand_3 && overlapping_exon == transcript.first_exon) || is_internal_tandem_duplication)) overlap_found = false;
		if (!overlap_found) return;
		if (imprecise_breakpoint) {
			gap = breakpoint + 1;
			positions[gap] = strand_3 ? a.exons[overlapping_exon].start : a.exons[overlapping_exon].end;
			sequence[gap] = strand_3 ? genome[positions[gap]] : complement_of(genome[positions[gap]]);
		}
		std::string from_assembly;
		std::vector<position_t> assembly_positions;
		for (int exon = overlapping_exon; exon != -1; exon = strand_3 ? a.exons[exon].next_exon : a.exons[exon].previous_exon) {
			const ExonRecord& e = a.exons[exon];
			for (position_t position = strand_3 ? std::max(e.start, positions[gap] + 1) : std::min(e.end, positions[gap] - 1); position >= e.start && position <= e.end; position += strand_3 ? +1 : -1) {
				from_assembly += strand_3 ? genome[position] : complement_of(genome[position]);
				assembly_positions.push_back(position);
			}
			if ((strand_3 && e.next_exon != -1) || (!strand_3 && e.previous_exon != -1)) { from_assembly += "___"; assembly_positions.resize(assembly_positions.size() + 3, -1); }
		}
		sequence.resize(gap + 1); sequence += "(";
		positions.resize(gap + 1); positions.push_back(-1);
		sequence += from_assembly;
		positions.insert(positions.end(), assembly_positions.begin(), assembly_positions.end());
		sequence += ")"; positions.push_back(-1);
		if (imprecise_breakpoint) { std::swap(sequence[breakpoint + 1], sequence[breakpoint + 2]); std::swap(positions[breakpoint + 1], positions[breakpoint + 2]); }
		if ((strand_3 && positions[positions.size() - 2] == last_exon.end) || (!strand_3 && positions[positions.size() - 2] == first_exon.start)) { sequence += "$"; positions.push_back(-1); }
	}
}

namespace {

// reference: dna_to_protein (source/annotate_protein_domains.cpp:163-192)
char amino_acid_of(const std::string& triplet) {
	std::string t = triplet;
	for (size_t i = 0; i < t.size(); ++i) t[i] = (char) toupper(t[i]);
	static const char* const table[] = { "GCA", "GCC", "GCG", "GCT", "TGC", "TGT", "GAC", "GAT", "GAA", "GAG", "TTC", "TTT", "GGA", "GGC", "GGG", "GGT", "CAC", "CAT", "ATA", "ATC", "ATT", "AAA", "AAG",
		"CTA", "CTC", "CTG", "CTT", "TTA", "TTG", "ATG", "AAC", "AAT", "CCA", "CCC", "CCG", "CCT", "CAA", "CAG", "CGA", "CGC", "CGG", "CGT", "AGA", "AGG", "TCA", "TCC", "TCG", "TCT", "AGC", "AGT",
		"ACA", "ACC", "ACG", "ACT", "GTA", "GTC", "GTG", "GTT", "TGG", "TAC", "TAT", "TAA", "TAG", "TGA" };
	static const char acids[] = "AAAACCDDEEFFGGGGHHIIIKKLLLLLLMNNPPPPQQRRRRRRSSSSSSTTTTVVVVWYY***";
	// the reference decides eight amino acids from the first two letters alone: the third may be anything (e.g. '?')
	static const char* const by_two[] = { "GC", "GG", "CT", "CC", "CG", "TC", "AC", "GT" };
	static const char two_acids[] = "AGLPRSTV";
	for (size_t k = 0; k < 8; ++k) if (t.compare(0, 2, by_two[k]) == 0) return two_acids[k];
	for (size_t k = 0; k < sizeof(table) / sizeof(table[0]); ++k) if (t == table[k]) return acids[k];
	return '?';
}

// reference: get_reading_frame (:213-261); -1 = none.  start_exon = the exon with the start codon (-1 = none)
int reading_frame_of(const TranscriptInput& in, const std::vector<position_t>& transcribed_bases, int from, int to, int transcript, contig_t contig, bool forward, int& start_exon) {
	const Annotation& a = in.annotation;
	start_exon = transcript < 0 ? -1 : (forward ? a.transcripts[transcript].first_exon : a.transcripts[transcript].last_exon);
	while (start_exon != -1 && a.exons[start_exon].coding_region_start == -1) start_exon = forward ? a.exons[start_exon].next_exon : a.exons[start_exon].previous_exon;
	if (start_exon == -1) return -1; // non-coding
	const std::string& genome = in.assembly.sequence.at(contig);
	const std::string first_codon = forward ? genome.substr(a.exons[start_exon].coding_region_start, 3) : reverse_complement(genome.substr(a.exons[start_exon].coding_region_end - 2, 3));
	if (first_codon != "ATG") return -1; // the annotation is wrong
	int reading_frame = -1, transcribed_coding_base = -1;
	for (int exon = start_exon; exon != -1 && a.exons[exon].coding_region_start != -1 && transcribed_coding_base == -1; exon = forward ? a.exons[exon].next_exon : a.exons[exon].previous_exon) {
		const ExonRecord& e = a.exons[exon];
		for (int position = from; position <= to && transcribed_coding_base == -1; position++)
			if (e.coding_region_start <= transcribed_bases[position] && e.coding_region_end >= transcribed_bases[position]) transcribed_coding_base = position;
		if (transcribed_coding_base == -1) reading_frame = (reading_frame + e.coding_region_end - e.coding_region_start + 1) % 3;
		else {
			reading_frame += forward ? transcribed_bases[transcribed_coding_base] - e.coding_region_start : e.coding_region_end - transcribed_bases[transcribed_coding_base];
			reading_frame = (reading_frame + 1) % 3;
		}
	}
	if (transcribed_coding_base == -1) return -1; // no coding region is transcribed
	for (int position = transcribed_coding_base - 1; position >= from; --position)
		if (transcribed_bases[position] != -1) reading_frame = reading_frame == 0 ? 2 : reading_frame - 1; // control characters and insertions do not count
	return reading_frame;
}

// reference: translate_reference_protein (:195-211): last base of every codon -> amino acid of the wild type
void reference_protein_of(const TranscriptInput& in, int start_exon, std::map<position_t, char>& protein) {
	if (start_exon == -1) return;
	const Annotation& a = in.annotation;
	const bool forward = a.genes[a.exons[start_exon].gene].strand;
	const std::string& genome = in.assembly.sequence.at(a.exons[start_exon].contig);
	std::string codon;
	bool warned = false;
	for (int exon = start_exon; exon != -1; exon = forward ? a.exons[exon].next_exon : a.exons[exon].previous_exon) {
		const ExonRecord& e = a.exons[exon];
		for (position_t position = forward ? e.coding_region_start : e.coding_region_end; position != -1 && position >= e.coding_region_start && position <= e.coding_region_end; position += forward ? +1 : -1) {
			codon += forward ? genome[position] : complement_of(genome[position]);
			if (codon.size() < 3) continue;
			protein[position] = amino_acid_of(codon);
			codon.clear();
			if (!warned && position < e.coding_region_end && position > e.coding_region_start && protein[position] == '*') {
				char text[512];
				snprintf(text, sizeof(text), "WARNING: encountered early stop codon in transcript %s at amino acid %zu (error in GTF file?) => predicted peptide sequence may be wrong\n", a.transcripts[e.transcript].name.c_str(), protein.size());
				if (transcript_warnings != NULL) *transcript_warnings += text; else fputs(text, stderr); // (rows are formatted by several threads: the writer prints the warnings in row order)
				warned = true;
			}
		}
	}
}

}

// reference: get_fusion_peptide_sequence (:263-393)
std::string fusion_peptide_sequence(const TranscriptInput& in, const std::string& sequence, const std::vector<position_t>& positions, const PeptideGenes& genes, int transcript_5, int transcript_3) {
	if (sequence.empty() || sequence == "." || sequence.find("...|") < sequence.size() || sequence.find("|...") < sequence.size()) return "."; // uncertain around the junction
	if (!in.assembly.has(genes.contig_5) || !in.assembly.has(genes.contig_3)) return ".";
	// the 5' part, possibly non-template bases, the 3' part; nothing beyond "..."
	const size_t end_5 = sequence.find('|') - 1;
	size_t start_5 = sequence.rfind("...", end_5);
	if (start_5 >= sequence.size()) start_5 = 0;
	else while (positions[start_5] == -1 && sequence[start_5] != '|') start_5++;
	size_t non_template_length = sequence.find('|', end_5 + 2);
	if (non_template_length >= sequence.size()) non_template_length = 0; else non_template_length -= end_5 + 2;
	size_t start_3 = end_5 + 2;
	if (non_template_length > 0) start_3 += non_template_length + 1;
	size_t end_3 = sequence.find("...", start_3);
	if (end_3 >= sequence.size()) end_3 = sequence.size() - 1; else end_3--;

	int start_exon_5 = -1, start_exon_3 = -1;
	int frame_5 = genes.dummy_5 ? -1 : reading_frame_of(in, positions, (int) start_5, (int) end_5, transcript_5, genes.contig_5, genes.forward_5, start_exon_5);
	if (frame_5 == -1) return "."; // no coding exon of the 5' gene is transcribed
	if (frame_5 != 0) frame_5 = 3 - frame_5;
	int frame_3 = -1;
	if (!genes.dummy_3 && genes.forward_3 == genes.predicted_strand_3) frame_3 = reading_frame_of(in, positions, (int) start_3, (int) end_3, transcript_3, genes.contig_3, genes.forward_3, start_exon_3); // not for anti-sense transcription
	std::map<position_t, char> protein_5, protein_3; // the wild type, to mark amino acids that differ
	reference_protein_of(in, start_exon_5, protein_5);
	reference_protein_of(in, start_exon_3, protein_3);

	const ExonRecord& first_coding_exon = in.annotation.exons[start_exon_5];
	std::string peptide, codon;
	int codon_5_bases = 0, codon_3_bases = 0; // does the codon span the breakpoint?
	bool started = false;
	for (size_t position = start_5 + frame_5; position < end_3; ++position) {
		if (!started) { // not before the start codon
			if (positions[position] != -1 && ((genes.forward_5 && positions[position] >= first_coding_exon.coding_region_start) || (!genes.forward_5 && positions[position] <= first_coding_exon.coding_region_end))) started = true;
			else continue;
		}
		const char base = sequence[position];
		if (base == 'A' || base == 'T' || base == 'C' || base == 'G' || base == 'a' || base == 't' || base == 'c' || base == 'g' || base == '?') {
			if (codon.empty()) codon_5_bases = codon_3_bases = 0;
			if (position <= end_5) codon_5_bases++; else if (position >= start_3) codon_3_bases++;
			codon += base;
		}
		if (codon.size() == 3) {
			char amino_acid = amino_acid_of(codon);
			const std::map<position_t, char>& wild_type = position <= end_5 ? protein_5 : protein_3;
			const std::map<position_t, char>::const_iterator expected = wild_type.find(positions[position]);
			if ((position > end_5 && position < start_3) ||                         // non-template bases
			    expected == wild_type.end() || amino_acid != expected->second ||    // differs from the wild type
			    (codon_5_bases != 3 && position <= end_5) || (codon_3_bases != 3 && position >= start_3) || // spans a breakpoint
			    (position >= start_3 && frame_3 == -1))                             // the 3' end is not coding
				amino_acid = (char) tolower(amino_acid);
			peptide += amino_acid;
			codon.clear();
			if (codon_3_bases >= 2 && amino_acid == '*') break; // a stop codon in the 3' gene
		}
		if ((position == end_5 && codon.size() <= 1) || (codon_5_bases == 2 && codon.empty())) // the end of the 5' part
			if (peptide.empty() || peptide[peptide.size() - 1] != '|') peptide += '|';
		if (non_template_length > 0 && ((position + 2 == start_3 && codon.size() <= 1) || (codon_3_bases == 1 && codon.empty()))) // the beginning of the 3' part
			if (peptide.empty() || peptide[peptide.size() - 1] != '|') peptide += '|';
	}
	return peptide.empty() ? "." : peptide;
}

// reference: is_in_frame (:395-446)
std::string reading_frame_verdict(const std::string& peptide) {
	if (peptide == "." || peptide.empty() || peptide[peptide.size() - 1] == '|') return "."; // nothing of the 3' gene
	auto in_frame_between = [&](size_t begin, size_t end) { for (size_t k = begin; k < end; ++k) if (peptide[k] >= 'A' && peptide[k] <= 'Z') return true; return false; };
	const size_t junction = peptide.rfind('|'), last_stop = peptide.rfind('*', junction);
	size_t start_codon_behind_stop = peptide.find('m', last_stop);
	if (start_codon_behind_stop >= junction) start_codon_behind_stop = peptide.find('M', last_stop);
	if (last_stop < junction && start_codon_behind_stop >= junction) return "stop-codon"; // a stop codon in front of the junction and no start codon behind it
	// in-frame amino acids before a stop codon need in-frame amino acids behind it, otherwise the breakpoint is probably in the 3' UTR
	if (last_stop < junction && in_frame_between(0, last_stop) && !in_frame_between(last_stop + 1, junction)) return "stop-codon";
	const bool in_frame_5 = in_frame_between(last_stop < junction ? last_stop + 1 : 0, junction), in_frame_3 = in_frame_between(junction + 1, peptide.size());
	return in_frame_5 && in_frame_3 ? "in-frame" : "out-of-frame";
}

}
