// arriba_amd/csrc/host/ingest.cpp -- BAM ingest: classify STAR records into chimeric fragments and pack
// them into the structure-of-arrays batch.  Restates the reference's read_chimeric_alignments
// (source/read_chimeric_alignments.cpp:19-773) for the standard workflow (-x only, STAR
// --chimOutType WithinBAM) and coverage_t::add_fragment (source/read_stats.cpp:161-266).
#include "arriba_host.h"

#include <sys/resource.h>
#include <sys/syscall.h>
#include <atomic>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <iostream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <sched.h>
#include <thread>
#include <zlib.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace arriba {

// ---- byte sources -------------------------------------------------------------------------------

namespace {

// One open stream serves every container: the first bytes are read once (sniffed for the format) and handed to the chosen source, so that
// the path may be a pipe or /dev/stdin (the reference's standard invocation is `STAR ... | arriba -x /dev/stdin`, run_arriba.sh:42;
// sam_open reads any path once, source/read_chimeric_alignments.cpp:563).
class PrefixedFile {
public:
	PrefixedFile(FILE* file, const uint8_t* prefix, size_t prefix_size): file_(file), prefix_(prefix, prefix + prefix_size), taken_(0) {}
	~PrefixedFile() { if (file_ != NULL && file_ != stdin) fclose(file_); }
	size_t read(uint8_t* buffer, size_t capacity) { // like fread: short only at the end of the stream
		size_t n = std::min(capacity, prefix_.size() - taken_);
		if (n > 0) { memcpy(buffer, &prefix_[taken_], n); taken_ += n; }
		while (n < capacity) {
			const size_t got = fread(buffer + n, 1, capacity - n, file_);
			if (got == 0) {
				if (ferror(file_)) throw std::runtime_error("failed to load alignments");
				break;
			}
			n += got;
		}
		return n;
	}
private:
	FILE* file_;
	std::vector<uint8_t> prefix_;
	size_t taken_;
};

// uncompressed BAM stream (magic BAM\1 at byte 0)
class RawSource: public ByteSource {
public:
	RawSource(FILE* file, const uint8_t* prefix, size_t prefix_size): input_(file, prefix, prefix_size) {}
	size_t read(uint8_t* buffer, size_t capacity) { return input_.read(buffer, capacity); }
private:
	PrefixedFile input_;
};

// plain gzip (RFC 1952, possibly several members) that is not BGZF: one inflate stream on the reader thread
class GzipSource: public ByteSource {
public:
	GzipSource(FILE* file, const uint8_t* prefix, size_t prefix_size): input_(file, prefix, prefix_size), in_(1u << 20), in_fill_(0), in_at_(0), open_(false), end_(false) { memset(&stream_, 0, sizeof(stream_)); }
	~GzipSource() { if (open_) inflateEnd(&stream_); }
	size_t read(uint8_t* buffer, size_t capacity) {
		size_t produced = 0;
		while (produced == 0 && capacity > 0) {
			if (in_at_ == in_fill_) {
				if (end_) break;
				in_fill_ = input_.read(&in_[0], in_.size()); in_at_ = 0;
				if (in_fill_ == 0) { end_ = true; if (open_) throw std::runtime_error("failed to load alignments"); break; } // truncated member
			}
			if (!open_) {
				if (inflateInit2(&stream_, 15 + 16) != Z_OK) throw std::runtime_error("failed to load alignments");
				open_ = true;
			}
			stream_.next_in = &in_[in_at_]; stream_.avail_in = (unsigned int) (in_fill_ - in_at_);
			stream_.next_out = buffer; stream_.avail_out = (unsigned int) std::min<size_t>(capacity, 1u << 30);
			const int status = inflate(&stream_, Z_NO_FLUSH);
			if (status != Z_OK && status != Z_STREAM_END && status != Z_BUF_ERROR) throw std::runtime_error("failed to load alignments");
			in_at_ = in_fill_ - stream_.avail_in;
			produced = std::min<size_t>(capacity, 1u << 30) - stream_.avail_out;
			if (status == Z_STREAM_END) { inflateEnd(&stream_); open_ = false; } // the next member, if any, starts a new stream
		}
		return produced;
	}
private:
	PrefixedFile input_;
	std::vector<uint8_t> in_;
	size_t in_fill_, in_at_;
	z_stream stream_;
	bool open_, end_;
};

class MemorySource: public ByteSource {
public:
	MemorySource(const uint8_t* data, size_t size): data_(data), size_(size), position_(0) {}
	size_t read(uint8_t* buffer, size_t capacity) {
		size_t n = std::min(capacity, size_ - position_);
		memcpy(buffer, data_ + position_, n);
		position_ += n;
		return n;
	}
private:
	const uint8_t* data_; size_t size_, position_;
};

}

ByteSource* open_memory_source(const uint8_t* data, size_t size) { return new MemorySource(data, size); }

// ---- record decoding ----------------------------------------------------------------------------

namespace {

const uint16_t BAM_FPAIRED = 1, BAM_FPROPER_PAIR = 2, BAM_FUNMAP = 4, BAM_FMUNMAP = 8, BAM_FREVERSE = 16, BAM_FREAD1 = 64, BAM_FSECONDARY = 256, BAM_FDUP = 1024, BAM_FSUPPLEMENTARY = 2048;
const char NT16[] = "=ACMGRSVTWYHKDBN";

inline uint32_t le32(const uint8_t* p) { return (uint32_t) p[0] | (uint32_t) p[1] << 8 | (uint32_t) p[2] << 16 | (uint32_t) p[3] << 24; }
inline uint16_t le16(const uint8_t* p) { return (uint16_t) (p[0] | p[1] << 8); }

struct Record { // one decoded BAM alignment record (SAMv1 section 4.2)
	int32_t tid, pos;
	uint16_t flag;
	int32_t l_seq;
	std::string qname;
	std::vector<uint32_t> cigar;
	std::vector<uint8_t> seq;  // 4-bit packed
	std::vector<uint8_t> aux;
	uint32_t n_cigar() const { return cigar.size(); }
	bool reverse() const { return flag & BAM_FREVERSE; }
	bool forward_strand() const { return !(flag & BAM_FREVERSE); } // strand_t FORWARD == true
	char base(int i) const { return NT16[(seq[i >> 1] >> ((~i & 1) << 2)) & 15]; }
	int qlen(uint32_t n) const { int l = 0; for (uint32_t i = 0; i < n; ++i) if (cigar_consumes_query(cigar_op(cigar[i]))) l += cigar_len(cigar[i]); return l; }
	int rlen(uint32_t n) const { int l = 0; for (uint32_t i = 0; i < n; ++i) if (cigar_consumes_reference(cigar_op(cigar[i]))) l += cigar_len(cigar[i]); return l; }
	int endpos() const { int l = (!(flag & BAM_FUNMAP) && !cigar.empty()) ? rlen(cigar.size()) : 1; return pos + (l == 0 ? 1 : l); }
	// aux lookup; returns pointer at the type byte or NULL
	const uint8_t* aux_get(char t0, char t1) const {
		const uint8_t* s = aux.data();
		const uint8_t* end = s + aux.size();
		while (s + 3 <= end) {
			const uint8_t* value = s + 2;
			size_t size;
			switch (*value) {
				case 'A': case 'c': case 'C': size = 2; break;
				case 's': case 'S': size = 3; break;
				case 'i': case 'I': case 'f': size = 5; break;
				case 'd': size = 9; break;
				case 'Z': case 'H': { const uint8_t* p = value + 1; while (p < end && *p) ++p; if (p >= end) return NULL; size = p - value + 1; break; }
				case 'B': {
					if (value + 6 > end) return NULL;
					size_t element = (value[1] == 'c' || value[1] == 'C') ? 1 : (value[1] == 's' || value[1] == 'S') ? 2 : 4;
					size = 6 + element * le32(value + 2);
					break;
				}
				default: return NULL;
			}
			if (value + size > end) return NULL;
			if (s[0] == (uint8_t) t0 && s[1] == (uint8_t) t1) return value;
			s = value + size;
		}
		return NULL;
	}
	static int64_t aux_to_int(const uint8_t* s) {
		switch (*s) {
			case 'c': return (int8_t) s[1];
			case 'C': return s[1];
			case 's': return (int16_t) le16(s + 1);
			case 'S': return le16(s + 1);
			case 'i': return (int32_t) le32(s + 1);
			case 'I': return le32(s + 1);
			default: return 0;
		}
	}
};

// decodes the block of one alignment record (p points behind block_size)
void decode_record(const uint8_t* p, uint32_t block_size, Record& r) {
	r.tid = (int32_t) le32(p);
	r.pos = (int32_t) le32(p + 4);
	uint32_t l_read_name = p[8];
	uint32_t n_cigar = le16(p + 12);
	r.flag = le16(p + 14);
	r.l_seq = (int32_t) le32(p + 16);
	const uint8_t* q = p + 32;
	if (r.l_seq < 0 || l_read_name == 0) throw std::runtime_error("failed to load alignments"); // untrusted input: sizes are checked before they are used
	size_t fixed = (size_t) l_read_name + 4 * (size_t) n_cigar + ((size_t) r.l_seq + 1) / 2 + (size_t) r.l_seq;
	if (32 + fixed > block_size) throw std::runtime_error("failed to load alignments");
	r.qname.assign((const char*) q, strnlen((const char*) q, l_read_name));
	q += l_read_name;
	r.cigar.resize(n_cigar);
	for (uint32_t i = 0; i < n_cigar; ++i) r.cigar[i] = le32(q + 4 * i);
	q += 4 * n_cigar;
	r.seq.assign(q, q + (r.l_seq + 1) / 2);
	q += (r.l_seq + 1) / 2 + r.l_seq;
	r.aux.assign(q, p + block_size);
}

class BamStream {
public:
	explicit BamStream(ByteSource& source): source_(source), begin_(0), end_(0), eof_(false) { buffer_.resize(8u << 20); }
	bool need(size_t n) { // make n bytes available at begin_
		while (end_ - begin_ < n) {
			if (eof_) return false;
			if (begin_ > 0) { memmove(&buffer_[0], &buffer_[begin_], end_ - begin_); end_ -= begin_; begin_ = 0; }
			if (buffer_.size() < n) buffer_.resize(n + (n >> 1));
			size_t got = source_.read(&buffer_[end_], buffer_.size() - end_);
			if (got == 0) eof_ = true;
			end_ += got;
		}
		return true;
	}
	const uint8_t* data() const { return &buffer_[begin_]; }
	void consume(size_t n) { begin_ += n; }
	size_t available() const { return end_ - begin_; }
	void read_header(std::vector<std::string>& target_names) {
		if (!need(12) || memcmp(data(), "BAM\1", 4) != 0) throw std::runtime_error("failed to read SAM header");
		uint32_t l_text = le32(data() + 4);
		consume(8);
		if (!need(l_text + 4)) throw std::runtime_error("failed to read SAM header");
		consume(l_text);
		uint32_t n_ref = le32(data());
		consume(4);
		for (uint32_t i = 0; i < n_ref; ++i) {
			if (!need(4)) throw std::runtime_error("failed to read SAM header");
			uint32_t l_name = le32(data());
			consume(4);
			if (!need(l_name + 4)) throw std::runtime_error("failed to read SAM header");
			target_names.push_back(std::string((const char*) data(), l_name > 0 ? l_name - 1 : 0));
			consume(l_name + 4);
		}
	}
	// the next record as raw bytes (block_size + block, valid until the next call); false at clean end of stream
	bool next_raw(const uint8_t*& record, uint32_t& block_size) {
		if (!need(4)) {
			if (available() == 0) return false;
			throw std::runtime_error("failed to load alignments");
		}
		block_size = le32(data());
		if (block_size < 32 || !need(4 + (size_t) block_size)) throw std::runtime_error("failed to load alignments");
		record = data();
		consume(4 + (size_t) block_size);
		return true;
	}
private:
	ByteSource& source_;
	std::vector<uint8_t> buffer_;
	size_t begin_, end_;
	bool eof_;
};

// ---- in-flight fragment table -------------------------------------------------------------------

struct Alignment { // reference: alignment_t, source/common.hpp:191-207
	bool supplementary = false, first_in_pair = false, strand = false;
	contig_t contig = 0;
	position_t start = 0, end = 0;
	std::vector<uint32_t> cigar;
	std::string sequence;
	unsigned int preclipping() const { uint32_t op = cigar_op(cigar.at(0)); return (op == CIGAR_S || op == CIGAR_H) ? cigar_len(cigar.at(0)) : 0; }
	unsigned int postclipping() const { uint32_t op = cigar_op(cigar.at(cigar.size() - 1)); return (op == CIGAR_S || op == CIGAR_H) ? cigar_len(cigar.at(cigar.size() - 1)) : 0; }
};
struct Fragment { // reference: mates_t, source/common.hpp:212-219
	std::vector<Alignment> alignments;
	bool single_end = false, duplicate = false;
};
typedef std::unordered_map<std::string, Fragment> fragment_table_t;

const unsigned char CLIP_NONE = 0, CLIP_START = 1, CLIP_END = 2;

std::string decode_sequence(const Record& r) {
	std::string sequence(r.l_seq, 'N');
	for (int i = 0; i < r.l_seq; ++i)
		sequence[i] = r.base(i);
	return sequence;
}

// reference: source/read_chimeric_alignments.cpp:50-91
void add_chimeric_alignment(Fragment& mates, const Record& r, bool is_supplementary = false, unsigned int cigar_index = 0, unsigned char clip = CLIP_NONE) {
	mates.single_end = !(r.flag & BAM_FPAIRED);
	mates.duplicate = mates.duplicate || (r.flag & BAM_FDUP);
	mates.alignments.resize(mates.alignments.size() + 1);
	Alignment& alignment = mates.alignments.back();
	alignment.strand = r.forward_strand();
	alignment.first_in_pair = r.flag & BAM_FREAD1;
	alignment.contig = r.tid;
	alignment.supplementary = is_supplementary;
	if (!is_supplementary)
		alignment.sequence = decode_sequence(r);
	if (clip == CLIP_START) {
		alignment.start = r.pos + r.rlen(cigar_index);
		alignment.end = r.endpos() - 1;
		uint32_t clip_type = cigar_op(r.cigar[0]) == CIGAR_H ? CIGAR_H : CIGAR_S;
		alignment.cigar.resize(r.n_cigar() - cigar_index + 1);
		alignment.cigar[0] = cigar_make(r.qlen(cigar_index), clip_type);
		for (unsigned int i = cigar_index; i < r.n_cigar(); ++i)
			alignment.cigar[i - cigar_index + 1] = r.cigar[i];
	} else if (clip == CLIP_END) {
		alignment.start = r.pos;
		alignment.end = r.pos + r.rlen(cigar_index + 1) - 1;
		uint32_t clip_type = cigar_op(r.cigar[r.n_cigar() - 1]) == CIGAR_H ? CIGAR_H : CIGAR_S;
		alignment.cigar.resize(cigar_index + 2);
		for (unsigned int i = 0; i <= cigar_index; ++i)
			alignment.cigar[i] = r.cigar[i];
		alignment.cigar[cigar_index + 1] = cigar_make(r.l_seq - r.qlen(cigar_index + 1), clip_type);
	} else {
		alignment.start = r.pos;
		alignment.end = r.endpos() - 1;
		alignment.cigar = r.cigar;
	}
}

// reference: source/read_chimeric_alignments.cpp:19-41
bool find_spanning_intron(const Record& r, position_t gene1_end, position_t gene2_start, unsigned int& cigar_index, position_t& read_pos) {
	if (r.n_cigar() < 3)
		return false;
	position_t before = r.pos, after;
	for (unsigned int i = 0; i < r.n_cigar(); ++i) {
		uint32_t op = r.cigar[i];
		unsigned int op_length = cigar_consumes_reference(cigar_op(op)) ? cigar_len(op) : 0;
		after = before + op_length;
		if (cigar_op(op) == CIGAR_N && (before <= gene1_end && after > gene1_end || before < gene2_start && after >= gene2_start)) {
			cigar_index = i;
			read_pos = r.qlen(i);
			return true;
		}
		before = after;
	}
	return false;
}

// reference: source/annotation.cpp:558-567
void get_boundaries_of_biggest_gene(const std::vector<uint32_t>& genes, const Annotation& annotation, position_t& start, position_t& end) {
	start = -1; end = -1;
	for (size_t g = 0; g < genes.size(); ++g) {
		const GeneRecord& gene = annotation.genes[genes[g]];
		if (start == -1 || start > gene.start) start = gene.start;
		if (end == -1 || end < gene.end) end = gene.end;
	}
}

// reference: source/read_chimeric_alignments.cpp:93-193
bool extract_read_through_alignment(fragment_table_t& fragments, const std::string& read_name, const Record* forward_mate, const Record* reverse_mate, const Annotation& annotation, const FlatIndex& gene_index) {
	if (!forward_mate->forward_strand())
		std::swap(forward_mate, reverse_mate);
	std::vector<uint32_t> forward_genes, reverse_genes, common_genes;
	if (forward_mate != NULL) get_annotation_by_coordinate(forward_mate->tid, forward_mate->pos, forward_mate->pos, forward_genes, gene_index);
	else get_annotation_by_coordinate(reverse_mate->tid, reverse_mate->pos, reverse_mate->pos, forward_genes, gene_index);
	if (reverse_mate != NULL) get_annotation_by_coordinate(reverse_mate->tid, reverse_mate->endpos(), reverse_mate->endpos(), reverse_genes, gene_index);
	else get_annotation_by_coordinate(forward_mate->tid, forward_mate->endpos(), forward_mate->endpos(), reverse_genes, gene_index);
	std::set_intersection(forward_genes.begin(), forward_genes.end(), reverse_genes.begin(), reverse_genes.end(), std::back_inserter(common_genes));
	if (!(common_genes.empty() && !(forward_genes.empty() && reverse_genes.empty())))
		return false;

	position_t forward_gene_start, forward_gene_end, reverse_gene_start, reverse_gene_end;
	get_boundaries_of_biggest_gene(forward_genes, annotation, forward_gene_start, forward_gene_end);
	get_boundaries_of_biggest_gene(reverse_genes, annotation, reverse_gene_start, reverse_gene_end);
	if (forward_gene_end == -1) forward_gene_end = reverse_gene_start - 1;
	if (reverse_gene_start == -1) reverse_gene_start = forward_gene_end + 1;

	unsigned int forward_cigar_op = 0, reverse_cigar_op = 0;
	position_t forward_read_pos = 0, reverse_read_pos = 0;
	bool forward_has_intron = (forward_mate == NULL) ? false : find_spanning_intron(*forward_mate, forward_gene_end, reverse_gene_start, forward_cigar_op, forward_read_pos);
	bool reverse_has_intron = (reverse_mate == NULL) ? false : find_spanning_intron(*reverse_mate, forward_gene_end, reverse_gene_start, reverse_cigar_op, reverse_read_pos);
	if (forward_has_intron && (!reverse_has_intron || forward_read_pos < reverse_mate->l_seq - reverse_read_pos)) {
		std::pair<fragment_table_t::iterator, bool> mates = fragments.insert(std::make_pair(read_name, Fragment()));
		if (mates.second) {
			add_chimeric_alignment(mates.first->second, *forward_mate, false, forward_cigar_op + 1, CLIP_START);
			add_chimeric_alignment(mates.first->second, *forward_mate, true, forward_cigar_op - 1, CLIP_END);
			if (reverse_mate != NULL) {
				if (reverse_has_intron) add_chimeric_alignment(mates.first->second, *reverse_mate, false, reverse_cigar_op + 1, CLIP_START);
				else add_chimeric_alignment(mates.first->second, *reverse_mate);
			}
			return true;
		}
	} else if (reverse_has_intron) {
		std::pair<fragment_table_t::iterator, bool> mates = fragments.insert(std::make_pair(read_name, Fragment()));
		if (mates.second) {
			add_chimeric_alignment(mates.first->second, *reverse_mate, true, reverse_cigar_op + 1, CLIP_START);
			add_chimeric_alignment(mates.first->second, *reverse_mate, false, reverse_cigar_op - 1, CLIP_END);
			if (forward_mate != NULL) {
				if (forward_has_intron) add_chimeric_alignment(mates.first->second, *forward_mate, false, forward_cigar_op - 1, CLIP_END);
				else add_chimeric_alignment(mates.first->second, *forward_mate);
			}
			return true;
		}
	} else if (forward_mate != NULL && reverse_mate != NULL && reverse_mate->pos >= reverse_gene_start && forward_mate->endpos() <= forward_gene_end) {
		std::pair<fragment_table_t::iterator, bool> mates = fragments.insert(std::make_pair(read_name, Fragment()));
		if (mates.second) {
			add_chimeric_alignment(mates.first->second, *forward_mate);
			add_chimeric_alignment(mates.first->second, *reverse_mate);
		}
		return true;
	}
	return false;
}

// reference: source/read_chimeric_alignments.cpp:197-211
bool clipped_sequence_is_adapter(const Record* mate1, const Record* mate2) {
	if (mate1 == NULL || mate2 == NULL)
		return false;
	if (mate1->pos == mate2->pos) {
		if (mate1->reverse() && cigar_op(mate1->cigar[0]) == CIGAR_S && !mate2->reverse() && cigar_op(mate2->cigar.back()) == CIGAR_S &&
		    cigar_len(mate1->cigar[0]) == cigar_len(mate2->cigar.back()))
			return true;
		if (mate2->reverse() && cigar_op(mate2->cigar[0]) == CIGAR_S && !mate1->reverse() && cigar_op(mate1->cigar.back()) == CIGAR_S &&
		    cigar_len(mate2->cigar[0]) == cigar_len(mate1->cigar.back()))
			return true;
	}
	return false;
}

// reference: source/read_chimeric_alignments.cpp:215-336 (integer types chosen to reproduce its mixed signed/unsigned arithmetic)
bool is_tandem_duplication(const Record* r, const Assembly& assembly, const unsigned int max_itd_length, Alignment& tandem) {
	const unsigned int min_clipped_length = 12, min_duplication_length = 9, max_duplication_length = max_itd_length;
	const unsigned int max_mismatches = 1, max_non_template_bases = 6, min_alignment_length = 15;
	if (r == NULL)
		return false;
	unsigned int clipped_length = 0, clipped_position = 0;
	bool clipped_start = true;
	int direction = +1, window_start = 0, window_end = 0, extended_read_start = 0;
	if (cigar_op(r->cigar[0]) == CIGAR_S && cigar_len(r->cigar[0]) >= min_clipped_length) {
		clipped_length = cigar_len(r->cigar[0]);
		clipped_position = 0;
		direction = -1;
		window_start = r->pos + min_duplication_length - clipped_length;
		window_end = r->pos + max_duplication_length - clipped_length;
		extended_read_start = r->pos - clipped_length;
		clipped_start = true;
	}
	if (cigar_op(r->cigar.back()) == CIGAR_S && cigar_len(r->cigar.back()) >= std::max(min_clipped_length, clipped_length)) {
		clipped_length = cigar_len(r->cigar.back());
		clipped_position = r->l_seq - clipped_length;
		direction = +1;
		window_start = r->endpos() - max_duplication_length;
		window_end = r->endpos() - min_duplication_length;
		extended_read_start = r->endpos();
		clipped_start = false;
	}
	if (clipped_length == 0)
		return false;
	if (r->tid < 0 || !assembly.has(r->tid))
		return false;
	const std::string& contig_sequence = assembly.sequence[r->tid];
	if (window_end + max_duplication_length + clipped_length + 1 >= contig_sequence.size() ||
	    window_start <= (int) (max_duplication_length + clipped_length + 1))
		return false;

	std::string clipped(clipped_length, 'N');
	for (unsigned int i = 0; i < clipped_length; ++i)
		clipped[i] = r->base(clipped_position + i);

	const float min_extended_align_fraction = 0.7;
	unsigned int extended_matches = 0;
	for (unsigned int read_pos = 0; read_pos < clipped_length; ++read_pos)
		if (extended_read_start + read_pos < contig_sequence.size()) // unsigned arithmetic as in the reference
			if (contig_sequence[extended_read_start + read_pos] == clipped[read_pos])
				extended_matches++;
	if (1.0 * extended_matches / clipped_length >= min_extended_align_fraction)
		return false;

	for (int contig_pos = window_start; contig_pos <= window_end; ++contig_pos) {
		unsigned int matches = 0, mismatches = 0;
		tandem.start = contig_sequence.size();
		tandem.end = -1;
		for (unsigned int i = 0; i < clipped_length; i++) {
			int read_pos = (direction == +1) ? i : clipped_length - 1 - i;
			if (contig_sequence[contig_pos + read_pos] == clipped[read_pos]) {
				matches++;
				if (contig_pos + read_pos < tandem.start) tandem.start = contig_pos + read_pos;
				if (contig_pos + read_pos > tandem.end) tandem.end = contig_pos + read_pos;
			} else if (i >= max_non_template_bases) {
				mismatches++;
				if (mismatches > max_mismatches)
					break;
			}
		}
		if (matches >= min_alignment_length || matches + mismatches == clipped_length) {
			tandem.strand = r->forward_strand();
			tandem.first_in_pair = r->flag & BAM_FREAD1;
			tandem.contig = r->tid;
			tandem.supplementary = !(r->flag & BAM_FPAIRED) || clipped_start && r->forward_strand() || !clipped_start && !r->forward_strand();
			if (!tandem.supplementary)
				tandem.sequence = decode_sequence(*r);
			uint32_t clip_left = clipped_start ? 0 : r->l_seq - clipped_length;
			uint32_t clip_right = clipped_start ? r->l_seq - clipped_length : 0;
			if (tandem.start > contig_pos) clip_left += tandem.start - contig_pos;
			if (tandem.end < contig_pos + (int) clipped_length - 1) clip_right += contig_pos + clipped_length - 1 - tandem.end;
			if (clip_left > 0) tandem.cigar.push_back(cigar_make(clip_left, CIGAR_S));
			tandem.cigar.push_back(cigar_make(tandem.end - tandem.start + 1, CIGAR_M));
			if (clip_right > 0) tandem.cigar.push_back(cigar_make(clip_right, CIGAR_S));
			return true;
		}
	}
	return false;
}

// reference: source/read_chimeric_alignments.cpp:340-373
bool disjoin_split_read_segments(Alignment& split_read, Alignment& supplementary) {
	const int min_remaining_supplementary_segment = 10;
	unsigned int clipped_split_read = split_read.strand ? split_read.preclipping() : split_read.postclipping();
	unsigned int clipped_supplementary = supplementary.strand ? supplementary.postclipping() : supplementary.preclipping();
	int overlap = (int) split_read.sequence.size() - clipped_split_read - clipped_supplementary;
	if (overlap <= 0)
		return true;
	unsigned int clipped_op = supplementary.strand ? supplementary.cigar.size() - 1 : 0;
	unsigned int matching_op = supplementary.strand ? clipped_op - 1 : 1;
	if (supplementary.cigar.size() < 2 || cigar_op(supplementary.cigar.at(matching_op)) != CIGAR_M ||
	    (int) cigar_len(supplementary.cigar.at(matching_op)) < overlap + min_remaining_supplementary_segment)
		return false;
	supplementary.cigar[clipped_op] = cigar_make(cigar_len(supplementary.cigar[clipped_op]) + overlap, cigar_op(supplementary.cigar[clipped_op]));
	supplementary.cigar[matching_op] = cigar_make(cigar_len(supplementary.cigar[matching_op]) - overlap, cigar_op(supplementary.cigar[matching_op]));
	if (supplementary.strand) supplementary.end -= overlap; else supplementary.start += overlap;
	return true;
}

inline bool complement_strand_if(bool strand, bool condition) { return condition ? !strand : strand; }

// reference: source/read_chimeric_alignments.cpp:377-506; returns false if the fragment is malformed
bool normalize_fragment(Fragment& f) {
	std::vector<Alignment>& a = f.alignments;
	if (f.single_end) {
		if (!(a.size() == 2 && (a[MATE1].supplementary != a[MATE2].supplementary)))
			return false;
		if (a[MATE1].end - a[MATE1].start > a[MATE2].end - a[MATE2].start) {
			a.push_back(a[MATE2]);
			a[MATE2] = a[MATE1];
		} else {
			a.push_back(a[MATE1]);
			a[MATE1] = a[MATE2];
		}
		if (!a[MATE1].supplementary) {
			a[SPLIT_READ].sequence = a[MATE1].sequence;
		} else if (!a[SPLIT_READ].supplementary) {
			a[MATE1].sequence = a[SPLIT_READ].sequence;
		} else {
			a[MATE1].sequence = a[SUPPLEMENTARY].sequence;
			a[SPLIT_READ].sequence = a[SUPPLEMENTARY].sequence;
		}
		a[SUPPLEMENTARY].sequence.clear();
		if (cigar_op(a[MATE1].cigar.at(0)) == CIGAR_H)
			a[MATE1].cigar[0] = cigar_make(cigar_len(a[MATE1].cigar[0]), CIGAR_S);
		if (cigar_op(a[MATE1].cigar.back()) == CIGAR_H)
			a[MATE1].cigar.back() = cigar_make(cigar_len(a[MATE1].cigar.back()), CIGAR_S);
		if (cigar_op(a[SPLIT_READ].cigar.at(0)) == CIGAR_H)
			a[SPLIT_READ].cigar[0] = cigar_make(cigar_len(a[SPLIT_READ].cigar[0]), CIGAR_S);
		if (cigar_op(a[SPLIT_READ].cigar.back()) == CIGAR_H) // the length is taken from MATE1's CIGAR at SPLIT_READ's last index, exactly as the reference does (:415)
			a[SPLIT_READ].cigar.back() = cigar_make(cigar_len(a[MATE1].cigar.at(a[SPLIT_READ].cigar.size() - 1)), CIGAR_S);
		a[SUPPLEMENTARY].supplementary = true;
		a[MATE1].supplementary = false;
		a[SPLIT_READ].supplementary = false;
		bool flip_mate1_strand;
		bool same = a[SPLIT_READ].strand == a[SUPPLEMENTARY].strand;
		if (a[SPLIT_READ].sequence.length() - a[SPLIT_READ].preclipping() - (same ? a[SUPPLEMENTARY].postclipping() : a[SUPPLEMENTARY].preclipping()) <
		    a[SPLIT_READ].sequence.length() - a[SPLIT_READ].postclipping() - (same ? a[SUPPLEMENTARY].preclipping() : a[SUPPLEMENTARY].postclipping()))
			flip_mate1_strand = a[SPLIT_READ].strand == true;
		else
			flip_mate1_strand = a[SPLIT_READ].strand == false;
		a[MATE1].strand = complement_strand_if(a[MATE1].strand, flip_mate1_strand);
		a[SPLIT_READ].strand = complement_strand_if(a[SPLIT_READ].strand, !flip_mate1_strand);
		a[SUPPLEMENTARY].strand = complement_strand_if(a[SUPPLEMENTARY].strand, !flip_mate1_strand);
		a[MATE1].first_in_pair = !flip_mate1_strand;
		a[SPLIT_READ].first_in_pair = flip_mate1_strand;
		a[SUPPLEMENTARY].first_in_pair = flip_mate1_strand;
		if (!disjoin_split_read_segments(a[SPLIT_READ], a[SUPPLEMENTARY]))
			return false;
	} else {
		if (a.size() == 3) {
			if (a[MATE1].supplementary) std::swap(a[MATE1], a[SUPPLEMENTARY]);
			else if (a[MATE2].supplementary) std::swap(a[MATE2], a[SUPPLEMENTARY]);
			if (a[SPLIT_READ].first_in_pair != a[SUPPLEMENTARY].first_in_pair)
				std::swap(a[MATE1], a[MATE2]);
			if (a[MATE1].supplementary || a[SPLIT_READ].supplementary || !a[SUPPLEMENTARY].supplementary)
				return false;
			if (a[MATE1].contig != a[SPLIT_READ].contig || a[MATE1].strand == a[SPLIT_READ].strand)
				return false;
			if (!disjoin_split_read_segments(a[SPLIT_READ], a[SUPPLEMENTARY]))
				return false;
		} else if (a.size() == 2) {
			if (a[MATE1].supplementary || a[MATE2].supplementary)
				return false;
		} else {
			return false;
		}
	}
	if (cigar_op(a[MATE1].cigar.at(0)) == CIGAR_H || cigar_op(a[MATE1].cigar.back()) == CIGAR_H ||
	    cigar_op(a[MATE2].cigar.at(0)) == CIGAR_H || cigar_op(a[MATE2].cigar.back()) == CIGAR_H)
		return false;
	return true;
}

// reference: source/read_chimeric_alignments.cpp:511-522
bool is_clipped_at_correct_end(const Record& r) {
	if (!(r.flag & BAM_FPAIRED))
		return true;
	unsigned int clipped_end;
	if (r.flag & BAM_FSUPPLEMENTARY)
		clipped_end = r.forward_strand() ? r.n_cigar() - 1 : 0;
	else
		clipped_end = r.forward_strand() ? 0 : r.n_cigar() - 1;
	uint32_t op = cigar_op(r.cigar[clipped_end]);
	return op == CIGAR_S || op == CIGAR_H;
}

// reference: source/read_chimeric_alignments.cpp:526-558
bool is_pristine_alignment(const Record& r) {
	for (unsigned int i = 0; i < r.n_cigar(); i++) {
		uint32_t op = cigar_op(r.cigar[i]);
		if (op != CIGAR_N && op != CIGAR_M && op != CIGAR_X)
			return false;
	}
	std::string sequence = decode_sequence(r);
	for (unsigned int i = 2, repeat = 0, count = 1; i + 2 < sequence.size(); i += 2) {
		if (sequence[i] == sequence[repeat] && sequence[i + 1] == sequence[repeat + 1]) {
			count++;
		} else if (sequence[i + 1] == sequence[repeat + 1] && sequence[i + 2] == sequence[repeat + 2]) {
			count++;
			i++;
		} else {
			count = 1;
			repeat = i;
		}
		if (count >= 8)
			return false;
	}
	return true;
}

// reference: source/read_stats.cpp:161-266.  flag1 is mate1's flag word as the caller left it (the
// reference zeroes it for discordant mates, source/read_chimeric_alignments.cpp:664).
// The coverage windows are shared by the ingest workers: +1 with saturation commutes, so relaxed atomics give the sequential result.
inline void increment_saturating(uint16_t& window) {
	uint16_t seen = __atomic_load_n(&window, __ATOMIC_RELAXED);
	while (seen < 65535 && !__atomic_compare_exchange_n(&window, &seen, (uint16_t) (seen + 1), true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
inline void set_flag(uint8_t& flag) { __atomic_store_n(&flag, (uint8_t) 1, __ATOMIC_RELAXED); }

void add_fragment_to_coverage(Coverage& coverage, const Record& mate1, uint16_t flag1, const Record* mate2_or_null, bool is_chimeric) {
	const Record& mate2 = (mate2_or_null == NULL) ? mate1 : *mate2_or_null;
	uint16_t flag2 = (mate2_or_null == NULL) ? flag1 : mate2.flag;
	if ((unsigned int) mate1.tid >= coverage.fragment_starts.size() || coverage.fragment_starts[mate1.tid].empty() ||
	    (unsigned int) mate2.tid >= coverage.fragment_starts.size() || coverage.fragment_starts[mate2.tid].empty())
		return;
	// the reference compares bam_cigar_type() (0..3) against BAM_CSOFT_CLIP (4), which never matches,
	// so only the proper-pair flag can turn a fragment chimeric here
	if ((flag1 & BAM_FPAIRED) && !(flag1 & BAM_FPROPER_PAIR))
		is_chimeric = true;
	if (!is_chimeric) {
		if (!(flag1 & BAM_FREVERSE) || !(flag1 & BAM_FPAIRED))
			set_flag(coverage.fragment_starts[mate1.tid][mate1.pos / COVERAGE_RESOLUTION]);
		else
			set_flag(coverage.fragment_starts[mate2.tid][mate2.pos / COVERAGE_RESOLUTION]);
	}
	position_t position1 = mate1.pos, position2 = mate2.pos;
	position_t position = std::min(position1, position2);
	int window = position / COVERAGE_RESOLUTION;
	unsigned int i1 = 0, i2 = 0;
	while (true) {
		uint32_t op1 = 0, op2 = 0;
		unsigned int length1, length2;
		if (i1 < mate1.n_cigar()) {
			op1 = mate1.cigar[i1];
			length1 = cigar_consumes_reference(cigar_op(op1)) ? cigar_len(op1) : 0;
		} else {
			length1 = 0;
			window = std::max(window, position2 / COVERAGE_RESOLUTION);
		}
		if (i2 < mate2.n_cigar()) {
			op2 = mate2.cigar[i2];
			length2 = cigar_consumes_reference(cigar_op(op2)) ? cigar_len(op2) : 0;
		} else {
			length2 = 0;
			window = std::max(window, position1 / COVERAGE_RESOLUTION);
		}
		int contig;
		uint32_t op;
		if (i1 < mate1.n_cigar() && (position1 + length1 < position2 + length2 || i2 >= mate2.n_cigar())) {
			i1++;
			if (length1 == 0) continue;
			op = op1; contig = mate1.tid; position1 += length1; position = position1;
		} else if (i2 < mate2.n_cigar()) {
			i2++;
			if (length2 == 0) continue;
			op = op2; contig = mate2.tid; position2 += length2; position = position2;
		} else {
			break;
		}
		std::vector<uint16_t>& windows = coverage.coverage[contig];
		if (cigar_consumes_query(cigar_op(op))) {
			while (window <= position / COVERAGE_RESOLUTION) {
				if (window >= 0 && (size_t) window < windows.size() && position - window * COVERAGE_RESOLUTION >= COVERAGE_RESOLUTION / 2)
					increment_saturating(windows[window]); // the reference tests `< 65535` first and then the half window: same result
				++window;
			}
		} else {
			window = position / COVERAGE_RESOLUTION;
		}
	}
	if (!is_chimeric) {
		if ((flag1 & BAM_FREVERSE) || !(flag1 & BAM_FPAIRED))
			set_flag(coverage.fragment_ends[mate1.tid][(position1 - 1) / COVERAGE_RESOLUTION]);
		else
			set_flag(coverage.fragment_ends[mate2.tid][(position2 - 1) / COVERAGE_RESOLUTION]);
	}
}

uint8_t encode_base(char c) {
	switch (c) {
		case '=': return 0; case 'A': return 1; case 'C': return 2; case 'M': return 3; case 'G': return 4; case 'R': return 5; case 'S': return 6; case 'V': return 7;
		case 'T': return 8; case 'W': return 9; case 'Y': return 10; case 'H': return 11; case 'K': return 12; case 'D': return 13; case 'B': return 14; default: return 15;
	}
}

// runs body(first, last) over [0, n) cut into one range per thread
template <class Body> void parallel_ranges(size_t n, unsigned int n_threads, const Body& body, size_t min_items = 4096) {
	if (n_threads <= 1 || n < min_items) { body((size_t) 0, n); return; }
	std::vector<std::thread> threads;
	for (unsigned int t = 0; t < n_threads; ++t)
		threads.push_back(std::thread([&body, n, n_threads, t] { worker_thread_starts(); body(n * t / n_threads, n * (t + 1) / n_threads); }));
	for (unsigned int t = 0; t < n_threads; ++t)
		threads[t].join();
}

// Two passes: the offsets of every fragment inside the pools (names, CIGARs, sequences) are a running sum over the fragments in name
// order; with them every fragment can be written independently, so the columns and pools are allocated once and filled by all threads.
void pack_batch(std::vector<std::pair<const std::string*, Fragment*> >& sorted, Batch& batch, unsigned int n_threads) {
	size_t n = sorted.size();
	batch.n = n;
	batch.n_aln.resize(n); batch.fbits.resize(n); batch.filter.assign(n, FILTER_none); batch.group.resize(n);
	for (int s = 0; s < 3; ++s) {
		batch.contig[s].assign(n, 0); batch.start[s].assign(n, 0); batch.end[s].assign(n, 0); batch.abits[s].assign(n, 0);
		batch.cigar_offset[s].assign(n, 0); batch.cigar_count[s].assign(n, 0);
	}
	for (int s = 0; s < 2; ++s) { batch.seq_offset[s].assign(n, 0); batch.seq_length[s].assign(n, 0); }
	batch.name_offset.assign(n + 1, 0);
	std::vector<size_t> cigar_base(n + 1, 0), seq_base(n + 1, 0); // seq_base in bytes
	std::vector<uint32_t> name_length(n), cigar_length(n), seq_bytes(n);
	std::vector<uint8_t> new_group(n, 0);
	parallel_ranges(n, n_threads, [&sorted, &name_length, &cigar_length, &seq_bytes, &new_group](size_t first, size_t last) {
		for (size_t i = first; i < last; ++i) {
			const std::string& name = *sorted[i].first;
			const Fragment& f = *sorted[i].second;
			name_length[i] = name.size();
			// multimapper groups: identical names up to the last ',' (source/common.hpp:222)
			if (i > 0) {
				const std::string& previous = *sorted[i - 1].first;
				size_t a = name.find_last_of(','), b = previous.find_last_of(',');
				new_group[i] = !(name.compare(0, a, previous, 0, b) == 0);
			}
			uint32_t cigar_words = 0, bytes = 0;
			for (size_t s = 0; s < f.alignments.size(); ++s) {
				cigar_words += f.alignments[s].cigar.size();
				if (s < 2) bytes += (((f.alignments[s].sequence.size() + 1) / 2) + 3) & ~(size_t) 3;
			}
			cigar_length[i] = cigar_words; seq_bytes[i] = bytes;
		}
	});
	size_t names_size = 0, cigar_size = 0, seq_size = 0;
	uint32_t group = 0;
	for (size_t i = 0; i < n; ++i) {
		batch.name_offset[i] = names_size; names_size += name_length[i];
		group += new_group[i]; batch.group[i] = group;
		cigar_base[i] = cigar_size; cigar_size += cigar_length[i];
		seq_base[i] = seq_size; seq_size += seq_bytes[i];
	}
	batch.name_offset[n] = names_size;
	if (names_size >= 0xFFFFFFFFull || cigar_size >= 0xFFFFFFFFull || seq_size / 4 >= 0xFFFFFFFFull)
		throw std::runtime_error("batch too large for 32-bit pool offsets");
	batch.names.assign(names_size, ' ');
	batch.cigar_pool.assign(cigar_size, 0);
	batch.seq_pool.assign(seq_size, 0);
	parallel_ranges(n, n_threads, [&sorted, &batch, &cigar_base, &seq_base](size_t first, size_t last) {
		for (size_t i = first; i < last; ++i) {
			const std::string& name = *sorted[i].first;
			const Fragment& f = *sorted[i].second;
			memcpy(&batch.names[batch.name_offset[i]], name.data(), name.size());
			batch.n_aln[i] = f.alignments.size();
			batch.fbits[i] = (f.single_end ? FBIT_SINGLE_END : 0) | (f.duplicate ? FBIT_DUPLICATE : 0);
			size_t cigar_at = cigar_base[i], seq_at = seq_base[i];
			for (size_t s = 0; s < f.alignments.size(); ++s) {
				const Alignment& a = f.alignments[s];
				batch.contig[s][i] = a.contig; batch.start[s][i] = a.start; batch.end[s][i] = a.end;
				batch.abits[s][i] = (a.strand ? ABIT_STRAND : 0) | (a.first_in_pair ? ABIT_FIRST_IN_PAIR : 0) | (a.supplementary ? ABIT_SUPPLEMENTARY : 0) | ABIT_PREDICTED_STRAND_AMBIGUOUS;
				batch.cigar_offset[s][i] = cigar_at;
				batch.cigar_count[s][i] = a.cigar.size();
				if (!a.cigar.empty()) memcpy(&batch.cigar_pool[cigar_at], a.cigar.data(), a.cigar.size() * sizeof(uint32_t));
				cigar_at += a.cigar.size();
				if (s < 2) {
					batch.seq_offset[s][i] = seq_at / 4;
					batch.seq_length[s][i] = a.sequence.size();
					for (size_t b = 0; b < a.sequence.size(); ++b)
						batch.seq_pool[seq_at + (b >> 1)] |= encode_base(a.sequence[b]) << ((~b & 1) << 2);
					seq_at += (((a.sequence.size() + 1) / 2) + 3) & ~(size_t) 3;
				}
			}
		}
	});
}

}

namespace {

// ---- the classification of the records (reference: the loop body of read_chimeric_alignments, source/read_chimeric_alignments.cpp:585-757) ----
// Everything a record does depends only on the records of the same read name that came before it (the parked first mate, the fragment's
// alignment list) and on shared read-only data; the counters and the coverage commute.  So the stream can be dealt to several workers by
// the hash of the read name: every worker sees the records of its names in stream order and keeps its own tables.
struct IngestShared {
	const Assembly& assembly; const Annotation& annotation; const FlatIndex& gene_index; const IngestOptions& options;
	std::vector<contig_t> tid_to_contig; std::vector<bool> interesting_tids, viral_contigs;
	Coverage& coverage;
};

struct IngestWorker {
	const IngestShared& shared;
	fragment_table_t fragments;
	std::unordered_map<std::string, Record> collated; // first mate parked until the second arrives
	bool no_chimeric_reads = true;
	uint64_t mapped_reads = 0;
	std::vector<uint64_t> mapped_viral_reads_by_contig;
	unsigned int malformed_count = 0, missing_hi_tag = 0;
	std::vector<std::pair<const std::string*, Fragment*> > sorted; // the valid fragments of this worker in name order (finish())
	std::string read_name;
	explicit IngestWorker(const IngestShared& shared_state, size_t n_contigs): shared(shared_state), mapped_viral_reads_by_contig(n_contigs, 0) {}

	void process(Record& record) {
		const IngestOptions& options = shared.options;
		if ((record.flag & BAM_FUNMAP) || (record.flag & BAM_FPAIRED) && (record.flag & BAM_FMUNMAP))
			return;
		int64_t hit_index = 1;
		const uint8_t* hi_tag = record.aux_get('H', 'I');
		if (hi_tag != NULL) {
			hit_index = Record::aux_to_int(hi_tag);
		} else if (record.flag & BAM_FSECONDARY) {
			missing_hi_tag++;
			return;
		}
		read_name = record.qname;
		read_name += "," + std::to_string(hit_index);
		if (record.tid < 0 || (size_t) record.tid >= shared.tid_to_contig.size())
			throw std::runtime_error("failed to load alignments");
		record.tid = shared.tid_to_contig[record.tid];

		if (record.flag & BAM_FSUPPLEMENTARY) {
			if (is_clipped_at_correct_end(record))
				add_chimeric_alignment(fragments[read_name], record, true);
			else
				malformed_count++;
			no_chimeric_reads = false;
			return;
		}
		if (shared.interesting_tids[record.tid])
			mapped_reads++;
		if ((record.flag & BAM_FPAIRED) && !(record.flag & BAM_FPROPER_PAIR)) {
			add_chimeric_alignment(fragments[read_name], record);
			no_chimeric_reads = false;
			if (!options.external_duplicate_marking || !(record.flag & BAM_FDUP))
				add_fragment_to_coverage(shared.coverage, record, 0 /* flag &= !BAM_FPAIRED zeroes the word */, NULL, true);
			return;
		}

		Record previous;
		bool have_previous = false;
		if (record.flag & BAM_FPAIRED) {
			std::unordered_map<std::string, Record>::iterator parked = collated.find(read_name);
			if (parked == collated.end()) {
				collated.emplace(read_name, std::move(record)); // (the caller decodes the next record into `record`: its buffers may be taken)
				return; // first mate: wait for the second
			}
			previous = std::move(parked->second);
			have_previous = true;
			collated.erase(parked);
		}
		const Record* previous_mate = have_previous ? &previous : NULL;

		bool is_tandem_alignment = false;
		Alignment tandem;
		if (!clipped_sequence_is_adapter(&record, previous_mate) &&
		    (previous_mate == NULL || record.forward_strand() != previous_mate->forward_strand()) &&
		    (is_tandem_duplication(&record, shared.assembly, options.max_itd_length, tandem) || is_tandem_duplication(previous_mate, shared.assembly, options.max_itd_length, tandem))) {
			Fragment& mates = fragments[read_name + "ITD"];
			add_chimeric_alignment(mates, record, record.forward_strand() == tandem.strand && !tandem.supplementary);
			if (previous_mate != NULL)
				add_chimeric_alignment(mates, *previous_mate, previous_mate->forward_strand() == tandem.strand && !tandem.supplementary);
			mates.alignments.push_back(tandem);
			is_tandem_alignment = true;
		}

		bool is_read_through = false;
		if (record.aux_get('S', 'A') != NULL && is_clipped_at_correct_end(record) ||
		    previous_mate != NULL && previous_mate->aux_get('S', 'A') != NULL && is_clipped_at_correct_end(*previous_mate)) {
			Fragment& mates = fragments[read_name];
			add_chimeric_alignment(mates, record);
			if (previous_mate != NULL)
				add_chimeric_alignment(mates, *previous_mate);
			no_chimeric_reads = false;
		} else if (!is_tandem_alignment) {
			is_read_through = extract_read_through_alignment(fragments, read_name, &record, previous_mate, shared.annotation, shared.gene_index);
			if (shared.viral_contigs[record.tid])
				for (const Record* mate = &record; mate != NULL; mate = (mate == previous_mate) ? NULL : previous_mate)
					if (is_pristine_alignment(*mate))
						mapped_viral_reads_by_contig[mate->tid]++;
		}
		if (!options.external_duplicate_marking || !(record.flag & BAM_FDUP))
			add_fragment_to_coverage(shared.coverage, record, record.flag, previous_mate, is_read_through);
	}

	// sanity check + slot normalisation, then the worker's fragments in name order (hazard H3: std::string order of "QNAME,HI")
	void finish() {
		collated.clear();
		sorted.reserve(fragments.size());
		for (fragment_table_t::iterator fragment = fragments.begin(); fragment != fragments.end(); ++fragment) {
			if (normalize_fragment(fragment->second))
				sorted.push_back(std::make_pair(&fragment->first, &fragment->second));
			else
				malformed_count++;
		}
		std::sort(sorted.begin(), sorted.end(), [](const std::pair<const std::string*, Fragment*>& a, const std::pair<const std::string*, Fragment*>& b) { return *a.first < *b.first; });
	}
};

// a bounded queue of byte chunks (concatenated raw records) from the reader to one worker
struct ChunkQueue {
	std::mutex mutex;
	std::condition_variable not_empty, not_full;
	std::deque<std::vector<uint8_t> > chunks;
	bool closed = false;
	void push(std::vector<uint8_t>&& chunk) {
		std::unique_lock<std::mutex> lock(mutex);
		not_full.wait(lock, [this] { return chunks.size() < 8; });
		chunks.push_back(std::move(chunk));
		not_empty.notify_one();
	}
	void close() { std::unique_lock<std::mutex> lock(mutex); closed = true; not_empty.notify_all(); }
	bool pop(std::vector<uint8_t>& chunk) {
		std::unique_lock<std::mutex> lock(mutex);
		not_empty.wait(lock, [this] { return !chunks.empty() || closed; });
		if (chunks.empty()) return false;
		chunk = std::move(chunks.front());
		chunks.pop_front();
		not_full.notify_one();
		return true;
	}
};

unsigned int ingest_threads() { // ARRIBA_INGEST_THREADS overrides; the reader is one more thread
	const char* setting = getenv("ARRIBA_INGEST_THREADS");
	if (setting != NULL && atoi(setting) > 0) return (unsigned int) atoi(setting);
	unsigned int cores = cpu_budget();
	return std::max(1u, std::min(48u, cores > 1 ? cores - 1 : 1u)); // one reader deals the records out: more workers than this wait for it
}

}

// ahost_limit_threads_of_this_thread: what a thread of a session that does two things at once (the feed of the next sample beside the stages of this one, the writer of the last
// file beside both) may use of the budget -- every decision about a number of threads that is made on that thread sees the smaller budget
static thread_local unsigned int g_thread_limit = 0;
void limit_threads_of_this_thread(unsigned int n) { g_thread_limit = n; }
void worker_thread_starts() {
	static const int nice_of_workers = [] { const char* setting = getenv("ARRIBA_WORKER_NICE"); const int value = setting ? atoi(setting) : 10; return value < 0 ? 0 : value > 19 ? 19 : value; }();
	if (nice_of_workers > 0) (void) setpriority(PRIO_PROCESS, (id_t) syscall(SYS_gettid), nice_of_workers); // (per thread on Linux; raising the nice value needs no privilege)
}
static unsigned int whole_cpu_budget();
unsigned int cpu_budget() { const unsigned int whole = whole_cpu_budget(); return g_thread_limit > 0 ? std::max(1u, std::min(whole, g_thread_limit)) : whole; }
static unsigned int whole_cpu_budget() {
	static const unsigned int budget = [] {
		unsigned int cores = std::max(1u, std::thread::hardware_concurrency());
		cpu_set_t set;
		if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int allowed = CPU_COUNT(&set); if (allowed > 0) cores = std::min(cores, (unsigned int) allowed); }
		double quota = 0; // CPUs per period
		if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota|max> <period>"
			char first[64]; long long period = 0;
			if (fscanf(f, "%63s %lld", first, &period) == 2 && strcmp(first, "max") != 0 && period > 0) quota = (double) atoll(first) / (double) period;
			fclose(f);
		} else {
			long long q = -1, period = 0; // cgroup v1
			if (FILE* a = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(a, "%lld", &q) != 1) q = -1; fclose(a); }
			if (FILE* b = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(b, "%lld", &period) != 1) period = 0; fclose(b); }
			if (q > 0 && period > 0) quota = (double) q / (double) period;
		}
		if (quota >= 1) cores = std::min(cores, (unsigned int) quota);
		if (const char* setting = getenv("ARRIBA_CPU_BUDGET")) if (atoi(setting) > 0) cores = (unsigned int) atoi(setting); // (for measurements)
		return std::max(1u, cores);
	}();
	return budget;
}

// BGZF (the container of BAM files: gzip members of at most 64 KiB, each with its compressed size in an extra field) read block-parallel:
// a batch of raw bytes is cut into blocks by their headers, the blocks are inflated and CRC-checked by all threads into their places in
// the output (the uncompressed size of a block is in its trailer), and the stream is served from there.  zlib's gz layer does the same
// work -- above all the CRC -- on the one thread that also cuts the stream into records; it limited the ingest to ~0.8 GB/s.
class BgzfSource: public ByteSource {
public:
	static bool is_bgzf_header(const uint8_t* header, size_t available) { return available >= 18 && block_size_from_header(header, available) > 0; }
	static size_t block_size(const uint8_t* header, size_t available) { return block_size_from_header(header, available); }
	// inflates one block (any deflate content) into `target` and checks size and CRC-32 of the trailer
	static bool inflate_block(const uint8_t* block, size_t raw_size, uint8_t* target, size_t out_size) {
		const size_t data_offset = 12 + (block[10] | (size_t) block[11] << 8);
		if (raw_size < data_offset + 8) return false;
		z_stream stream;
		memset(&stream, 0, sizeof(stream));
		if (inflateInit2(&stream, -15) != Z_OK) return false;
		uint8_t nothing = 0;
		stream.next_in = const_cast<uint8_t*>(block + data_offset); stream.avail_in = (unsigned int) (raw_size - data_offset - 8);
		stream.next_out = target != NULL ? target : &nothing; stream.avail_out = (unsigned int) out_size;
		const int status = inflate(&stream, Z_FINISH);
		const bool complete = status == Z_STREAM_END && stream.total_out == out_size;
		inflateEnd(&stream);
		return complete && (uint32_t) crc32(crc32(0L, Z_NULL, 0), target != NULL ? target : &nothing, (unsigned int) out_size) == le32(block + raw_size - 8);
	}
	BgzfSource(FILE* file, const uint8_t* prefix, size_t prefix_size, unsigned int n_threads): input_(file, prefix, prefix_size), n_threads_(std::max(1u, n_threads)), raw_fill_(0), served_(0), end_of_file_(false) {
		raw_.resize(32u << 20);
	}
	size_t read(uint8_t* buffer, size_t capacity) {
		while (served_ == decoded_.size()) { // decode the next batch of blocks
			if (end_of_file_ && raw_fill_ == 0) return 0;
			decode_batch();
		}
		const size_t n = std::min(capacity, decoded_.size() - served_);
		memcpy(buffer, &decoded_[served_], n);
		served_ += n;
		return n;
	}
private:
	// total size of the block that starts with `header`, 0 if it is not a BGZF block header (RFC 1952 + the "BC" extra subfield of the SAM specification)
	static size_t block_size_from_header(const uint8_t* header, size_t available) {
		if (available < 12 || header[0] != 31 || header[1] != 139 || header[2] != 8 || !(header[3] & 4)) return 0;
		const size_t extra_length = header[10] | (size_t) header[11] << 8;
		if (available < 12 + extra_length) return 0;
		for (size_t at = 12; at + 4 <= 12 + extra_length; ) {
			const size_t subfield_length = header[at + 2] | (size_t) header[at + 3] << 8;
			if (header[at] == 'B' && header[at + 1] == 'C' && subfield_length == 2 && at + 6 <= 12 + extra_length) return (size_t) (header[at + 4] | (size_t) header[at + 5] << 8) + 1;
			at += 4 + subfield_length;
		}
		return 0;
	}
	struct Block { size_t raw_offset, raw_size, out_offset, out_size; };
	void decode_batch() {
		if (!end_of_file_) {
			const size_t got = input_.read(&raw_[raw_fill_], raw_.size() - raw_fill_);
			raw_fill_ += got;
			if (got == 0) end_of_file_ = true;
		}
		std::vector<Block> blocks;
		size_t at = 0, out = 0;
		while (raw_fill_ - at >= 18) {
			const size_t size = block_size_from_header(&raw_[at], raw_fill_ - at);
			if (size == 0 || size < 26) throw std::runtime_error("failed to load alignments");
			if (raw_fill_ - at < size) break; // the rest of the block comes with the next batch
			Block block = { at, size, out, (size_t) le32(&raw_[at + size - 4]) };
			blocks.push_back(block);
			at += size; out += block.out_size;
		}
		if (blocks.empty() && end_of_file_ && raw_fill_ > 0) throw std::runtime_error("failed to load alignments"); // a truncated block
		decoded_.resize(out);
		served_ = 0;
		std::vector<uint8_t> failed(n_threads_ + 1, 0);
		const std::vector<uint8_t>& raw = raw_; std::vector<uint8_t>& decoded = decoded_;
		parallel_ranges(blocks.size(), n_threads_, [&blocks, &raw, &decoded, &failed, this](size_t first, size_t last) {
			for (size_t b = first; b < last; ++b) {
				const Block& block = blocks[b];
				const uint8_t* header = &raw[block.raw_offset];
				const size_t data_offset = 12 + (header[10] | (size_t) header[11] << 8);
				if (block.raw_size < data_offset + 8) { failed[0] = 1; continue; }
				uint8_t* target = block.out_size > 0 ? &decoded[block.out_offset] : NULL;
				z_stream stream;
				memset(&stream, 0, sizeof(stream));
				if (inflateInit2(&stream, -15) != Z_OK) { failed[0] = 1; continue; }
				uint8_t nothing = 0;
				stream.next_in = const_cast<uint8_t*>(header + data_offset); stream.avail_in = (unsigned int) (block.raw_size - data_offset - 8);
				stream.next_out = target != NULL ? target : &nothing; stream.avail_out = (unsigned int) block.out_size;
				const int status = inflate(&stream, Z_FINISH);
				const bool complete = status == Z_STREAM_END && stream.total_out == block.out_size;
				inflateEnd(&stream);
				if (!complete || (uint32_t) crc32(crc32(0L, Z_NULL, 0), target != NULL ? target : &nothing, (unsigned int) block.out_size) != le32(header + block.raw_size - 8)) failed[0] = 1;
			}
		}, 8);
		if (failed[0]) throw std::runtime_error("failed to load alignments");
		memmove(&raw_[0], &raw_[at], raw_fill_ - at); // an incomplete block stays for the next batch
		raw_fill_ -= at;
	}
	PrefixedFile input_;
	unsigned int n_threads_;
	std::vector<uint8_t> raw_, decoded_;
	size_t raw_fill_, served_;
	bool end_of_file_;
};

// The container is recognised from the first 18 bytes of the ONE open stream (never by opening the path a second time: it may be a pipe):
// a BGZF block header -> block-parallel inflate; any other gzip member -> streaming inflate; otherwise the bytes are taken as the BAM stream.
ByteSource* open_bam_file(const std::string& path) {
	FILE* file = (path == "-") ? stdin : fopen(path.c_str(), "rb"); // htslib reads standard input for "-"
	if (file == NULL) throw std::runtime_error("failed to open SAM file");
	uint8_t header[18];
	size_t got = 0;
	while (got < sizeof(header)) {
		const size_t n = fread(header + got, 1, sizeof(header) - got, file);
		if (n == 0) break;
		got += n;
	}
	if (BgzfSource::is_bgzf_header(header, got)) return new BgzfSource(file, header, got, ingest_threads());
	if (got >= 2 && header[0] == 31 && header[1] == 139) return new GzipSource(file, header, got);
	return new RawSource(file, header, got);
}

// reference: source/read_chimeric_alignments.cpp:560-773
void read_chimeric_alignments(ByteSource& source, const Assembly& assembly, Contigs& contigs, const Annotation& annotation, const FlatIndex& gene_index, const IngestOptions& options, IngestResult& result) {
	BamStream stream(source);
	std::vector<std::string> target_names;
	stream.read_header(target_names);

	IngestShared shared = { assembly, annotation, gene_index, options, std::vector<contig_t>(target_names.size()), std::vector<bool>(target_names.size()), std::vector<bool>(), result.coverage };
	for (size_t target = 0; target < target_names.size(); ++target) {
		std::string contig_name = remove_chr(target_names[target]);
		shared.tid_to_contig[target] = contigs.add(target_names[target]);
		if (shared.tid_to_contig[target] >= shared.interesting_tids.size())
			shared.interesting_tids.resize(shared.tid_to_contig[target] + 1);
		shared.interesting_tids[shared.tid_to_contig[target]] = is_interesting_contig(contig_name, options.interesting_contigs);
	}
	result.coverage.resize(contigs, assembly);
	for (std::map<std::string, contig_t>::const_iterator contig = contigs.by_name.begin(); contig != contigs.by_name.end(); ++contig)
		if (!assembly.has(contig->second) && is_interesting_contig(contig->first, options.interesting_contigs))
			throw std::runtime_error("could not find sequence of contig '" + contig->first + "'");
	shared.viral_contigs.resize(contigs.size());
	for (std::map<std::string, contig_t>::const_iterator contig = contigs.by_name.begin(); contig != contigs.by_name.end(); ++contig)
		shared.viral_contigs[contig->second] = is_interesting_contig(contig->first, options.viral_contigs);

	const unsigned int n_workers = ingest_threads();
	const bool timing = getenv("ARRIBA_INGEST_TIMING") != NULL;
	const std::chrono::steady_clock::time_point started = std::chrono::steady_clock::now();
	auto lap = [&started, timing](const char* phase) { if (timing) std::cerr << "ingest: " << phase << " at " << std::chrono::duration<double>(std::chrono::steady_clock::now() - started).count() << " s" << std::endl; };
	std::vector<std::unique_ptr<IngestWorker> > workers;
	for (unsigned int w = 0; w < n_workers; ++w)
		workers.push_back(std::unique_ptr<IngestWorker>(new IngestWorker(shared, contigs.size())));

	const uint8_t* raw;
	uint32_t block_size;
	if (n_workers == 1) {
		Record record;
		while (stream.next_raw(raw, block_size)) {
			result.records++;
			decode_record(raw + 4, block_size, record);
			workers[0]->process(record);
		}
		workers[0]->finish();
	} else {
		// the reader only cuts the stream into records and deals them out by the hash of the read name; the workers decode and classify
		std::vector<std::unique_ptr<ChunkQueue> > queues;
		std::vector<std::exception_ptr> failures(n_workers);
		std::vector<std::thread> threads;
		for (unsigned int w = 0; w < n_workers; ++w)
			queues.push_back(std::unique_ptr<ChunkQueue>(new ChunkQueue()));
		for (unsigned int w = 0; w < n_workers; ++w)
			threads.push_back(std::thread([w, &workers, &queues, &failures] {
				std::vector<uint8_t> chunk;
				Record record;
				bool failed = false;
				while (queues[w]->pop(chunk)) {
					if (failed) continue; // keep draining so that the reader never blocks
					try {
						for (size_t at = 0; at < chunk.size(); ) {
							uint32_t size = le32(&chunk[at]);
							decode_record(&chunk[at + 4], size, record);
							workers[w]->process(record);
							at += 4 + (size_t) size;
						}
					} catch (...) { failures[w] = std::current_exception(); failed = true; }
				}
				if (!failed) {
					try { workers[w]->finish(); } catch (...) { failures[w] = std::current_exception(); }
				}
			}));
		const size_t chunk_bytes = 1u << 20;
		std::vector<std::vector<uint8_t> > filling(n_workers);
		std::exception_ptr reader_failure;
		try {
			while (stream.next_raw(raw, block_size)) {
				result.records++;
				const uint32_t l_read_name = raw[4 + 8];
				if (36 + (size_t) l_read_name > 4 + (size_t) block_size) throw std::runtime_error("failed to load alignments");
				uint64_t hash = 1469598103934665603ull; // FNV-1a over the read name
				for (const uint8_t* c = raw + 36; c < raw + 36 + l_read_name && *c; ++c) hash = (hash ^ *c) * 1099511628211ull;
				std::vector<uint8_t>& chunk = filling[(hash >> 17) % n_workers];
				if (chunk.capacity() == 0) chunk.reserve(chunk_bytes + (64u << 10));
				chunk.insert(chunk.end(), raw, raw + 4 + (size_t) block_size);
				if (chunk.size() >= chunk_bytes) {
					const unsigned int w = (unsigned int) (&chunk - &filling[0]);
					queues[w]->push(std::move(chunk));
					chunk = std::vector<uint8_t>();
				}
			}
		} catch (...) { reader_failure = std::current_exception(); }
		lap("stream cut into records and dealt out");
		for (unsigned int w = 0; w < n_workers; ++w) {
			if (!filling[w].empty() && !reader_failure) queues[w]->push(std::move(filling[w]));
			queues[w]->close();
		}
		for (unsigned int w = 0; w < n_workers; ++w)
			threads[w].join();
		if (reader_failure) std::rethrow_exception(reader_failure);
		for (unsigned int w = 0; w < n_workers; ++w)
			if (failures[w]) std::rethrow_exception(failures[w]);
	}

	lap("records classified, fragments normalised and sorted per worker");
	bool no_chimeric_reads = true;
	result.mapped_viral_reads_by_contig.assign(contigs.size(), 0);
	size_t n_fragments = 0;
	for (unsigned int w = 0; w < n_workers; ++w) {
		const IngestWorker& worker = *workers[w];
		result.mapped_reads += worker.mapped_reads;
		result.malformed_count += worker.malformed_count;
		result.missing_hi_tag += worker.missing_hi_tag;
		no_chimeric_reads = no_chimeric_reads && worker.no_chimeric_reads;
		for (size_t contig = 0; contig < worker.mapped_viral_reads_by_contig.size(); ++contig)
			result.mapped_viral_reads_by_contig[contig] += worker.mapped_viral_reads_by_contig[contig];
		n_fragments += worker.sorted.size();
	}
	if (result.mapped_reads == 0)
		throw std::runtime_error("no normal reads found");
	if (result.malformed_count > 0)
		std::cerr << "WARNING: " << result.malformed_count << " SAM records were malformed and ignored" << std::endl;
	if (no_chimeric_reads)
		throw std::runtime_error("no split reads or discordant mates found (STAR must either be run with '--chimOutType WithinBAM' or the file 'Chimeric.out.sam' must be passed to Arriba via the argument -c)");
	if (result.missing_hi_tag > 0)
		std::cerr << "WARNING: " << result.missing_hi_tag << " secondary alignments lack the 'HI' tag and were ignored (STAR must be run with '--outSAMattributes HI' for Arriba to make use of multi-mapping reads for fusion detection)" << std::endl;

	// merge of the workers' name-sorted lists (names are unique across workers)
	std::vector<std::pair<const std::string*, Fragment*> > sorted;
	sorted.reserve(n_fragments);
	if (n_workers == 1) {
		sorted.swap(workers[0]->sorted);
	} else {
		typedef std::pair<const std::string*, Fragment*> Entry;
		std::vector<size_t> cursor(n_workers, 0);
		auto later = [&workers, &cursor](unsigned int a, unsigned int b) { return *workers[b]->sorted[cursor[b]].first < *workers[a]->sorted[cursor[a]].first; };
		std::vector<unsigned int> heap;
		for (unsigned int w = 0; w < n_workers; ++w)
			if (!workers[w]->sorted.empty()) heap.push_back(w);
		std::make_heap(heap.begin(), heap.end(), later);
		while (!heap.empty()) {
			std::pop_heap(heap.begin(), heap.end(), later);
			const unsigned int w = heap.back();
			const Entry& entry = workers[w]->sorted[cursor[w]];
			sorted.push_back(entry);
			if (++cursor[w] < workers[w]->sorted.size()) std::push_heap(heap.begin(), heap.end(), later);
			else heap.pop_back();
		}
	}
	lap("lists merged");
	pack_batch(sorted, result.batch, n_workers);
	lap("batch packed");
	// the tables are millions of heap nodes: every worker's table is freed by a thread of its own
	parallel_ranges(n_workers, n_workers > 1 ? n_workers : 0, [&workers](size_t first, size_t last) { for (size_t w = first; w < last; ++w) workers[w].reset(); }, 2);
	lap("tables freed");
}


// ---- feeding the device ingest (agpu_ingest_*: read_chimeric_alignments on the GPU) ----------------------------------------------------------------
// The host opens the file once, parses the BAM header (the contigs of the run must be known before the records are classified) and hands the bytes on
// in pieces: a BGZF file whose blocks are stored (STAR --outBAMcompression 0, run_arriba.sh:34) goes to the device as it is, with the table of
// its blocks -- the payloads are moved into place in HBM; deflated blocks are inflated here by all cores straight into the caller's (pinned) buffer.

namespace {

// reads n bytes at the current position of a descriptor: all threads with pread on a regular file, one read loop on a pipe
struct FileBytes {
	int fd; bool seekable; uint64_t position; unsigned int n_threads;
	uint64_t limit = ~(uint64_t) 0; // reads stop here (a part of a file: BamFeed::take_part)
	size_t read(uint8_t* buffer, size_t capacity) {
		if (limit != ~(uint64_t) 0) capacity = position >= limit ? 0 : (size_t) std::min<uint64_t>(capacity, limit - position);
		if (capacity == 0) return 0;
		if (!seekable || capacity < (8u << 20) || n_threads <= 1) {
			size_t got = 0;
			while (got < capacity) {
				const ssize_t n = seekable ? pread(fd, buffer + got, capacity - got, (off_t) (position + got)) : ::read(fd, buffer + got, capacity - got);
				if (n < 0) { if (errno == EINTR) continue; throw std::runtime_error("failed to load alignments"); }
				if (n == 0) break;
				got += (size_t) n;
			}
			position += got;
			return got;
		}
		std::atomic<size_t> got(0);
		std::atomic<bool> failed(false);
		const uint64_t base = position;
		parallel_ranges(capacity, n_threads, [&](size_t first, size_t last) {
			size_t done = 0;
			while (first + done < last) {
				const ssize_t n = pread(fd, buffer + first + done, last - first - done, (off_t) (base + first + done));
				if (n < 0) { if (errno == EINTR) continue; failed = true; break; }
				if (n == 0) break;
				done += (size_t) n;
			}
			got += done;
		}, 1);
		if (failed) throw std::runtime_error("failed to load alignments");
		const size_t total = got; // the bytes form a prefix: a short range means the end of the file
		position += total;
		return total;
	}
};

}

class BamFeed {
public:
	enum Mode { RAW, BGZF_STORED, BGZF_DEFLATED, GZIP };
	BamFeed(const std::string& path): gzip_open_(false), end_(false), n_targets_(0), header_size_(0), header_raw_end_(0), consumed_raw_(0), head_block_raw_(NOWHERE), head_skip_(0), tail_block_raw_(NOWHERE), tail_keep_(0) {
		fd_ = (path == "-") ? 0 : open(path.c_str(), O_RDONLY);
		if (fd_ < 0) throw std::runtime_error("failed to open SAM file");
		file_.fd = fd_; file_.position = 0; file_.n_threads = std::min(32u, cpu_budget());
		if (const char* knob = getenv("ARRIBA_FEED_THREADS")) if (atoi(knob) > 0) file_.n_threads = (unsigned int) std::min(atoi(knob), 256); // (how many threads read a piece of the file: for measurements)
		struct stat status;
		file_.seekable = fstat(fd_, &status) == 0 && S_ISREG(status.st_mode);
		file_size_ = file_.seekable ? (uint64_t) status.st_size : 0;
		n_threads_ = ingest_threads();
		// the first bytes decide the container; they stay in `pending_` and are delivered again as the start of the stream
		pending_.resize(1u << 20);
		pending_.resize(file_.read(&pending_[0], pending_.size()));
		if (BgzfSource::is_bgzf_header(pending_.data(), pending_.size())) mode_ = block_is_stored(pending_.data(), pending_.size()) ? BGZF_STORED : BGZF_DEFLATED;
		else if (pending_.size() >= 2 && pending_[0] == 31 && pending_[1] == 139) mode_ = GZIP;
		else mode_ = RAW;
	}
	~BamFeed() { if (gzip_open_) inflateEnd(&gzip_); if (fd_ > 0) close(fd_); }

	// the BAM header (magic, text, reference names), from the start of the uncompressed stream; returns its size
	uint64_t read_header(std::vector<std::string>& target_names) {
		std::vector<uint8_t> head;
		size_t consumed_raw = 0; // BGZF: bytes of pending_ already inflated into head
		while (true) {
			if (mode_ == RAW) head = pending_;
			else if (mode_ == GZIP) { head.clear(); inflate_prefix(head); }
			else {
				while (true) {
					const size_t block = BgzfSource::block_size(pending_.data() + consumed_raw, pending_.size() - consumed_raw);
					if (block == 0 || pending_.size() - consumed_raw < block) break;
					const size_t out = le32(&pending_[consumed_raw + block - 4]), at = head.size();
					head.resize(at + out);
					if (!BgzfSource::inflate_block(&pending_[consumed_raw], block, out > 0 ? &head[at] : NULL, out)) throw std::runtime_error("failed to read SAM header");
					consumed_raw += block;
				}
			}
			uint64_t size = 0;
			if (parse_header(head, target_names, size)) {
				n_targets_ = (uint32_t) target_names.size(); header_size_ = size;
				// BGZF: the file offset of the block that holds the first record, and where that record starts inside it
				if (mode_ == BGZF_STORED || mode_ == BGZF_DEFLATED) {
					size_t raw_at = 0; uint64_t out_at = 0;
					while (true) {
						const size_t block = BgzfSource::block_size(pending_.data() + raw_at, pending_.size() - raw_at);
						if (block == 0 || pending_.size() - raw_at < block) break;
						const uint64_t out = le32(&pending_[raw_at + block - 4]);
						if (out_at + out > size) break;
						raw_at += block; out_at += out;
					}
					header_raw_end_ = raw_at; header_inside_ = size - out_at;
				}
				return size;
			}
			// the header is longer than what is there: read more
			const size_t before = pending_.size();
			pending_.resize(before + (4u << 20));
			const size_t got = file_.read(&pending_[before], pending_.size() - before);
			pending_.resize(before + got);
			if (got == 0) throw std::runtime_error("failed to read SAM header");
		}
	}
	uint64_t stream_size_hint() const { return (mode_ == RAW || mode_ == BGZF_STORED) ? file_size_ + (1u << 20) : 0; }

	// the next piece; false at the end of the file.  kind 1: `buffer` holds raw BGZF bytes whose blocks are all stored, `blocks` their table
	bool next(uint8_t* buffer, size_t capacity, agpu_bgzf_block* blocks, uint32_t block_capacity, ahost_bam_piece& piece) {
		memset(&piece, 0, sizeof(piece));
		if (capacity < (1u << 20)) throw std::runtime_error("piece buffer too small");
		if (mode_ == RAW) {
			size_t n = take_pending(buffer, capacity);
			if (n < capacity && !end_) { const size_t got = file_.read(buffer + n, capacity - n); if (got == 0) end_ = true; n += got; }
			piece.bytes = n; piece.stream_bytes = n;
			consumed_raw_ += n;
			return n > 0;
		}
		if (mode_ == GZIP) {
			const size_t n = inflate_stream(buffer, capacity);
			piece.bytes = n; piece.stream_bytes = n;
			return n > 0;
		}
		if (mode_ == BGZF_STORED) {
			size_t n = take_pending(buffer, capacity);
			if (n < capacity && !end_) { const size_t got = file_.read(buffer + n, capacity - n); if (got == 0) end_ = true; n += got; }
			if (n == 0) return false;
			size_t at = 0, out = 0; uint32_t count = 0;
			bool deflated_ahead = false;
			while (n - at >= 18) {
				const size_t size = BgzfSource::block_size(buffer + at, n - at);
				if (size == 0 || size < 26) throw std::runtime_error("failed to load alignments");
				if (n - at < size) break;
				if (!block_is_stored(buffer + at, size)) { deflated_ahead = true; break; }
				if (count == block_capacity) break;
				const size_t data_offset = 12 + (buffer[at + 10] | (size_t) buffer[at + 11] << 8), payload = le32(buffer + at + size - 4);
				size_t lo = 0, hi = payload; // (the first and the last block of a part of the file give only the records of the part)
				if (consumed_raw_ + at == head_block_raw_) lo = std::min<size_t>(head_skip_, payload);
				if (consumed_raw_ + at == tail_block_raw_) hi = std::max<size_t>(lo, std::min<size_t>(tail_keep_, payload));
				if (hi > lo) {
					agpu_bgzf_block& b = blocks[count++];
					b.raw_offset = at; b.payload_offset = (uint32_t) (data_offset + 5 + lo); b.payload_size = (uint32_t) (hi - lo); b.stream_offset = out; b.crc32 = (lo == 0 && hi == payload) ? le32(buffer + at + size - 8) : 0; b.isize = 0; b.skip = 0; b.keep = 0;
				}
				at += size; out += hi - lo;
			}
			if (at == 0 && !deflated_ahead && end_) throw std::runtime_error("failed to load alignments"); // a truncated block at the end of the file
			pending_.assign(buffer + at, buffer + n); // an incomplete block (or everything from the first deflated block on) waits for the next call
			if (deflated_ahead) mode_ = BGZF_DEFLATED;
			consumed_raw_ += at;
			piece.stored_bgzf = 1; piece.bytes = at; piece.stream_bytes = out; piece.n_blocks = count;
			return at > 0 || deflated_ahead || !pending_.empty();
		}
		// BGZF_DEFLATED: the blocks go to the device as they are, with their table (piece.stored_bgzf = 2): bgzf_inflate_kernel makes the stream in HBM, a quarter of the bytes
		// cross the link.  ARRIBA_HOST_INFLATE=1: inflated here by all threads instead (the way of rounds 2-3, kept for measurements and as the second implementation in the tests)
		static const bool host_inflate = getenv("ARRIBA_HOST_INFLATE") != NULL && getenv("ARRIBA_HOST_INFLATE")[0] == '1';
		if (!host_inflate) {
			size_t n = take_pending(buffer, capacity);
			if (n < capacity && !end_) { const size_t got = file_.read(buffer + n, capacity - n); if (got == 0) end_ = true; n += got; }
			if (n == 0) return false;
			size_t at = 0, out = 0; uint32_t count = 0;
			const size_t out_limit = (size_t) 3 << 30; // (of one piece: the stream grows by that much at once)
			while (n - at >= 18) {
				const size_t size = BgzfSource::block_size(buffer + at, n - at);
				if (size == 0 || size < 26) throw std::runtime_error("failed to load alignments");
				if (n - at < size) break;
				if (count == block_capacity) break;
				const size_t data_offset = 12 + (buffer[at + 10] | (size_t) buffer[at + 11] << 8), out_size = le32(buffer + at + size - 4);
				if (size < data_offset + 8 || out_size > 65536) throw std::runtime_error("failed to load alignments");
				if (out + out_size > out_limit) break;
				size_t lo = 0, hi = out_size; // (the first and the last block of a part of the file give only the records of the part)
				if (consumed_raw_ + at == head_block_raw_) lo = std::min<size_t>(head_skip_, out_size);
				if (consumed_raw_ + at == tail_block_raw_) hi = std::max<size_t>(lo, std::min<size_t>(tail_keep_, out_size));
				if (hi > lo) {
					agpu_bgzf_block& b = blocks[count++];
					b.raw_offset = at; b.payload_offset = (uint32_t) data_offset; b.payload_size = (uint32_t) (size - data_offset - 8); b.stream_offset = out; b.crc32 = (lo == 0 && hi == out_size) ? le32(buffer + at + size - 8) : 0;
					b.isize = (uint32_t) out_size; b.skip = (uint32_t) lo; b.keep = (uint32_t) (hi - lo);
				}
				at += size; out += hi - lo;
			}
			if (at == 0 && end_) throw std::runtime_error("failed to load alignments"); // a truncated block at the end of the file
			pending_.assign(buffer + at, buffer + n);
			consumed_raw_ += at;
			piece.stored_bgzf = 2; piece.bytes = at; piece.stream_bytes = out; piece.n_blocks = count;
			return at > 0 || !pending_.empty();
		}
		raw_.resize(std::max<size_t>(capacity / 3, 4u << 20));
		size_t n = take_pending(raw_.data(), raw_.size());
		if (n < raw_.size() && !end_) { const size_t got = file_.read(&raw_[n], raw_.size() - n); if (got == 0) end_ = true; n += got; }
		if (n == 0) return false;
		struct Block { size_t raw_offset, raw_size, out_offset, out_size, lo, hi; };
		std::vector<Block> list;
		size_t at = 0, out = 0;
		while (n - at >= 18) {
			const size_t size = BgzfSource::block_size(&raw_[at], n - at);
			if (size == 0 || size < 26) throw std::runtime_error("failed to load alignments");
			if (n - at < size) break;
			const size_t out_size = le32(&raw_[at + size - 4]);
			if (out + out_size > capacity) break;
			size_t lo = 0, hi = out_size; // (the first and the last block of a part of the file give only the records of the part)
			if (consumed_raw_ + at == head_block_raw_) lo = std::min<size_t>(head_skip_, out_size);
			if (consumed_raw_ + at == tail_block_raw_) hi = std::max<size_t>(lo, std::min<size_t>(tail_keep_, out_size));
			Block block = { at, size, out, out_size, lo, hi };
			list.push_back(block);
			at += size; out += hi - lo;
		}
		if (list.empty() && end_) throw std::runtime_error("failed to load alignments");
		std::vector<uint8_t> failed(1, 0);
		const std::vector<uint8_t>& raw = raw_;
		parallel_ranges(list.size(), n_threads_, [&list, &raw, buffer, &failed](size_t first, size_t last) {
			for (size_t b = first; b < last; ++b) {
				const Block& block = list[b];
				if (block.lo == 0 && block.hi == block.out_size) {
					if (!BgzfSource::inflate_block(&raw[block.raw_offset], block.raw_size, block.out_size > 0 ? buffer + block.out_offset : NULL, block.out_size)) failed[0] = 1;
				} else {
					std::vector<uint8_t> whole(block.out_size + 1);
					if (!BgzfSource::inflate_block(&raw[block.raw_offset], block.raw_size, whole.data(), block.out_size)) failed[0] = 1;
					else memcpy(buffer + block.out_offset, whole.data() + block.lo, block.hi - block.lo);
				}
			}
		}, 8);
		if (failed[0]) throw std::runtime_error("failed to load alignments");
		pending_.assign(raw_.begin() + at, raw_.begin() + n);
		consumed_raw_ += at;
		piece.bytes = out; piece.stream_bytes = out;
		return !list.empty() || !pending_.empty();
	}
	// ---- a part of the file (one sample over several GPUs: every rank reads the records of its part; include/arriba_host.h: ahost_bam_open_part) ----------
	// The file is cut where a new read name begins (the alignments of a read follow each other in STAR's output, as the reference needs them for its
	// single pass only within a name: source/read_chimeric_alignments.cpp:596-749), near the byte offsets size * k / parts.  cut(target) is a function of
	// the file alone -- the first record start at or behind `target` that a chain of plausible records confirms, then on to the first record with another
	// name -- so the rank before the cut and the rank behind it find the same place without talking to each other.
	// Returns the offset of the first record of the part in the stream this feed delivers (the header size for part 0, else 0).
	uint64_t take_part(uint32_t part, uint32_t parts) {
		if (parts == 0 || part >= parts) throw std::runtime_error("part of the sample out of range");
		if (!file_.seekable || mode_ == GZIP) throw std::runtime_error("a part of a sample can only be read from a BAM file on disk (BGZF or uncompressed): every rank opens the file at its own offset");
		if (header_size_ == 0) throw std::runtime_error("the BAM header must be read first");
		// told before any rank reads a byte of its part (the check of the read names behind the exchange of the parts would find it, too -- after every rank has ingested its part)
		if (parts > 1 && sorted_by_coordinate_) throw std::runtime_error("the BAM file is sorted by coordinate (@HD SO:coordinate): its alignments of one read are apart, and a sample is read in parts by several GPUs "
		                                                                  "only if they follow each other (STAR's output order; `samtools collate` otherwise) -- or run it on one GPU, which collates by name itself");
		const Cut begin = part == 0 ? Cut() : cut(file_size_ / parts * part), end = part + 1 == parts ? Cut() : cut(file_size_ / parts * (part + 1));
		if (part > 0) {
			pending_.clear(); end_ = false;
			file_.position = begin.at_end ? file_size_ : begin.raw; consumed_raw_ = file_.position;
			if (mode_ != RAW && !begin.at_end && begin.inside > 0) { head_block_raw_ = begin.raw; head_skip_ = begin.inside; }
			if (mode_ == BGZF_DEFLATED || mode_ == BGZF_STORED) mode_ = BGZF_STORED; // (a block that is not stored switches to the inflating path by itself)
		}
		if (part + 1 < parts && !end.at_end) {
			uint64_t stop = end.raw;
			if (mode_ != RAW && end.inside > 0) { tail_block_raw_ = end.raw; tail_keep_ = end.inside; stop = end.raw_next; }
			if (stop < consumed_raw_) stop = consumed_raw_; // (cuts are monotone in their targets; an empty part if they coincide)
			file_.limit = stop;
			if (part == 0 && pending_.size() > stop) pending_.resize(stop); // the sniffed first bytes already reach beyond the part
			if (mode_ != RAW && head_block_raw_ == tail_block_raw_ && head_block_raw_ != NOWHERE && tail_keep_ < head_skip_) tail_keep_ = head_skip_;
		}
		return part == 0 ? header_size_ : 0;
	}
private:
	static const uint64_t NOWHERE = ~(uint64_t) 0;
	struct Cut { bool at_end; uint64_t raw, inside, raw_next; Cut(): at_end(true), raw(0), inside(0), raw_next(0) {} }; // RAW: raw = offset of the record; BGZF: raw = offset of its block, inside = offset in the block's payload, raw_next = the block behind
	struct Window { std::vector<uint8_t> bytes; std::vector<uint64_t> block_raw, block_out; uint64_t raw_begin; bool reaches_end; }; // uncompressed bytes from a block boundary on; per block: file offset, offset in `bytes`

	size_t pread_at(uint64_t offset, uint8_t* buffer, size_t capacity) const {
		size_t got = 0;
		while (got < capacity) {
			const ssize_t n = pread(fd_, buffer + got, capacity - got, (off_t) (offset + got));
			if (n < 0) { if (errno == EINTR) continue; throw std::runtime_error("failed to load alignments"); }
			if (n == 0) break;
			got += (size_t) n;
		}
		return got;
	}
	// a record that could be one: sizes that add up, reference ids of the header, a printable nul-terminated name
	bool plausible_record(const std::vector<uint8_t>& w, uint64_t at, uint64_t& next) const {
		if (at + 36 > w.size()) return false;
		const uint64_t block_size = le32(&w[at]);
		const int32_t ref = (int32_t) le32(&w[at + 4]), pos = (int32_t) le32(&w[at + 8]), next_ref = (int32_t) le32(&w[at + 24]), next_pos = (int32_t) le32(&w[at + 28]), l_seq = (int32_t) le32(&w[at + 20]);
		const uint32_t l_read_name = w[at + 12], n_cigar = w[at + 16] | (uint32_t) w[at + 17] << 8;
		if (block_size < 32 + 2 || block_size > (64u << 20) || l_read_name < 2 || l_seq < 0 || ref < -1 || ref >= (int32_t) n_targets_ || next_ref < -1 || next_ref >= (int32_t) n_targets_ || pos < -1 || next_pos < -1) return false;
		if (32 + (uint64_t) l_read_name + 4 * (uint64_t) n_cigar + ((uint64_t) l_seq + 1) / 2 + (uint64_t) l_seq > block_size) return false;
		if (at + 36 + l_read_name > w.size()) return false;
		for (uint32_t k = 0; k + 1 < l_read_name; ++k) if (w[at + 36 + k] < 33 || w[at + 36 + k] > 126) return false;
		if (w[at + 36 + l_read_name - 1] != 0) return false;
		next = at + 4 + block_size;
		return true;
	}
	// CHAIN records in a row (or fewer, when they end exactly where the file ends)
	bool chain_starts_at(const Window& w, uint64_t at) const {
		const int CHAIN = 32;
		for (int k = 0; k < CHAIN; ++k) {
			if (at == w.bytes.size() && w.reaches_end) return k > 0;
			uint64_t next;
			if (!plausible_record(w.bytes, at, next)) return false;
			if (next > w.bytes.size()) return false;
			at = next;
		}
		return true;
	}
	// the uncompressed bytes of the file from `raw_begin` (a block boundary; any byte of an uncompressed file) on, at least `want` of them unless the file ends
	void load_window(uint64_t raw_begin, uint64_t want, Window& w) const {
		w.bytes.clear(); w.block_raw.clear(); w.block_out.clear(); w.raw_begin = raw_begin; w.reaches_end = false;
		if (mode_ == RAW) {
			w.bytes.resize(want);
			w.bytes.resize(pread_at(raw_begin, w.bytes.data(), want));
			w.reaches_end = raw_begin + w.bytes.size() >= file_size_;
			return;
		}
		uint64_t raw_at = raw_begin;
		std::vector<uint8_t> block(1u << 16);
		while (w.bytes.size() < want) {
			if (raw_at >= file_size_) { w.reaches_end = true; break; }
			const size_t got = pread_at(raw_at, block.data(), block.size());
			const size_t size = BgzfSource::block_size(block.data(), got);
			if (size < 26 || got < size) throw std::runtime_error("failed to load alignments");
			const size_t out = le32(&block[size - 4]), at = w.bytes.size();
			w.block_raw.push_back(raw_at); w.block_out.push_back(at);
			w.bytes.resize(at + out);
			if (!BgzfSource::inflate_block(block.data(), size, out > 0 ? &w.bytes[at] : NULL, out)) throw std::runtime_error("failed to load alignments");
			raw_at += size;
		}
		w.block_raw.push_back(raw_at); w.block_out.push_back(w.bytes.size()); // (one behind the last block: the end)
	}
	// the first block boundary at or behind `target`: a BGZF header that the headers of the two blocks behind it confirm
	uint64_t block_boundary(uint64_t target) const {
		std::vector<uint8_t> bytes(256u << 10);
		while (true) {
			if (target >= file_size_) return file_size_;
			const size_t got = pread_at(target, bytes.data(), bytes.size());
			for (size_t i = 0; i + 18 <= got; ++i) {
				size_t at = i; int confirmed = 0; bool valid = true;
				while (confirmed < 3) {
					if (target + at == file_size_ || at + 18 > got) break; // the chain ends with the file / with what was read
					const size_t size = BgzfSource::block_size(&bytes[at], got - at);
					if (size < 26) { valid = false; break; }
					at += size; ++confirmed;
				}
				if (valid && confirmed > 0 && target + at <= file_size_) return target + i;
			}
			if (got < bytes.size()) return file_size_;
			target += got - 18;
		}
	}
	Cut cut(uint64_t target) const {
		Cut result;
		uint64_t raw_begin, search_from = 0;
		if (mode_ == RAW) raw_begin = std::max<uint64_t>(target, header_size_);
		else {
			raw_begin = block_boundary(target);
			if (raw_begin <= header_raw_end_) { raw_begin = header_raw_end_; search_from = header_inside_; } // not before the first record
		}
		if (raw_begin >= file_size_) return result;
		for (uint64_t want = 4u << 20; ; want *= 4) {
			Window w;
			load_window(raw_begin, want, w);
			// the first record start a chain confirms
			uint64_t first = NOWHERE;
			for (uint64_t i = search_from; i + 36 <= w.bytes.size(); ++i) if (chain_starts_at(w, i)) { first = i; break; }
			if (first == NOWHERE) {
				if (w.reaches_end) return result; // no record behind the target: the part behind this cut is empty
				if (want >= (1u << 30)) throw std::runtime_error("failed to load alignments");
				continue;
			}
			// on to the first record with another name
			uint64_t at = first, next = 0; bool found = false, ran_out = false;
			const uint32_t name_length = w.bytes[first + 12];
			while (true) {
				if (at == w.bytes.size() && w.reaches_end) break; // the name of the cut goes on to the end of the file
				if (!plausible_record(w.bytes, at, next) || next > w.bytes.size()) { ran_out = true; break; }
				if (at != first && (w.bytes[at + 12] != name_length || memcmp(&w.bytes[at + 36], &w.bytes[first + 36], name_length) != 0)) { found = true; break; }
				at = next;
			}
			if (ran_out) {
				if (w.reaches_end || want >= (1u << 30)) throw std::runtime_error("failed to load alignments");
				continue;
			}
			if (!found) return result;
			result.at_end = false;
			if (mode_ == RAW) { result.raw = raw_begin + at; return result; }
			size_t block = 0;
			while (block + 2 < w.block_out.size() && w.block_out[block + 1] <= at) ++block;
			result.raw = w.block_raw[block]; result.inside = at - w.block_out[block]; result.raw_next = w.block_raw[block + 1];
			return result;
		}
	}
	static bool block_is_stored(const uint8_t* block, size_t available) { // one stored deflate block that fills the member
		const size_t size = BgzfSource::block_size(block, available);
		if (size == 0 || available < size) return false;
		const size_t data_offset = 12 + (block[10] | (size_t) block[11] << 8);
		if (size < data_offset + 5 + 8) return false;
		const size_t payload = le32(block + size - 4), length = block[data_offset + 1] | (size_t) block[data_offset + 2] << 8, inverse = block[data_offset + 3] | (size_t) block[data_offset + 4] << 8;
		return block[data_offset] == 1 && length == payload && (length ^ inverse) == 0xFFFF && data_offset + 5 + payload + 8 == size;
	}
	bool parse_header(const std::vector<uint8_t>& head, std::vector<std::string>& target_names, uint64_t& size) {
		target_names.clear();
		if (head.size() < 12) return false;
		if (memcmp(head.data(), "BAM\1", 4) != 0) throw std::runtime_error("failed to read SAM header");
		uint64_t at = 8 + (uint64_t) le32(&head[4]);
		if (head.size() < at + 4) return false;
		{ // the sort order the header declares ("@HD ... SO:coordinate"): a file sorted by coordinate keeps the alignments of a read apart
			const std::string text((const char*) &head[8], (size_t) le32(&head[4]));
			const size_t line_end = text.find('\n');
			const std::string first_line = text.substr(0, line_end);
			sorted_by_coordinate_ = first_line.compare(0, 3, "@HD") == 0 && first_line.find("SO:coordinate") != std::string::npos;
		}
		const uint32_t n_ref = le32(&head[at]);
		at += 4;
		for (uint32_t i = 0; i < n_ref; ++i) {
			if (head.size() < at + 4) return false;
			const uint32_t l_name = le32(&head[at]);
			at += 4;
			if (head.size() < at + l_name + 4) return false;
			target_names.push_back(std::string((const char*) &head[at], l_name > 0 ? l_name - 1 : 0));
			at += (uint64_t) l_name + 4;
		}
		size = at;
		return true;
	}
	size_t take_pending(uint8_t* buffer, size_t capacity) {
		const size_t n = std::min(capacity, pending_.size());
		if (n > 0) { memcpy(buffer, pending_.data(), n); pending_.erase(pending_.begin(), pending_.begin() + n); }
		return n;
	}
	// plain gzip: the header is parsed from a throw-away inflate of the bytes read so far, the stream is inflated again from the start when it is delivered
	void inflate_prefix(std::vector<uint8_t>& head) {
		z_stream stream; memset(&stream, 0, sizeof(stream));
		if (inflateInit2(&stream, 15 + 16) != Z_OK) throw std::runtime_error("failed to read SAM header");
		head.resize(std::max<size_t>(pending_.size() * 8, 1u << 20));
		stream.next_in = pending_.data(); stream.avail_in = (unsigned int) pending_.size(); stream.next_out = head.data(); stream.avail_out = (unsigned int) head.size();
		const int status = inflate(&stream, Z_SYNC_FLUSH);
		head.resize(stream.total_out);
		inflateEnd(&stream);
		if (status != Z_OK && status != Z_STREAM_END && status != Z_BUF_ERROR) throw std::runtime_error("failed to read SAM header");
	}
	size_t inflate_stream(uint8_t* buffer, size_t capacity) {
		size_t produced = 0;
		while (produced < capacity) {
			if (pending_.empty()) {
				if (end_) break;
				pending_.resize(4u << 20);
				const size_t got = file_.read(&pending_[0], pending_.size());
				pending_.resize(got);
				if (got == 0) { end_ = true; if (gzip_open_) throw std::runtime_error("failed to load alignments"); break; }
			}
			if (!gzip_open_) { memset(&gzip_, 0, sizeof(gzip_)); if (inflateInit2(&gzip_, 15 + 16) != Z_OK) throw std::runtime_error("failed to load alignments"); gzip_open_ = true; }
			gzip_.next_in = pending_.data(); gzip_.avail_in = (unsigned int) pending_.size();
			gzip_.next_out = buffer + produced; gzip_.avail_out = (unsigned int) std::min<size_t>(capacity - produced, 1u << 30);
			const size_t room = gzip_.avail_out;
			const int status = inflate(&gzip_, Z_NO_FLUSH);
			if (status != Z_OK && status != Z_STREAM_END && status != Z_BUF_ERROR) throw std::runtime_error("failed to load alignments");
			produced += room - gzip_.avail_out;
			pending_.erase(pending_.begin(), pending_.end() - gzip_.avail_in);
			if (status == Z_STREAM_END) { inflateEnd(&gzip_); gzip_open_ = false; }
		}
		return produced;
	}
	int fd_;
	FileBytes file_;
	uint64_t file_size_;
	unsigned int n_threads_;
	Mode mode_;
	std::vector<uint8_t> pending_, raw_;
	z_stream gzip_;
	bool gzip_open_, end_;
	bool sorted_by_coordinate_ = false; // the header says SO:coordinate
	uint32_t n_targets_;
	uint64_t header_size_, header_raw_end_, header_inside_ = 0; // BGZF: file offset of the block that holds the first record, and the record's offset inside it
	uint64_t consumed_raw_;                                    // file offset of the first byte the next piece starts with
	uint64_t head_block_raw_, head_skip_, tail_block_raw_, tail_keep_; // a part of the file: payload bytes to skip in its first block / to keep of its last one
};

BamFeed* open_bam_feed(const std::string& path) { return new BamFeed(path); }
void close_bam_feed(BamFeed* feed) { delete feed; }
uint64_t bam_feed_header(BamFeed* feed, std::vector<std::string>& target_names) { return feed->read_header(target_names); }
uint64_t bam_feed_size_hint(BamFeed* feed) { return feed->stream_size_hint(); }
uint64_t bam_feed_take_part(BamFeed* feed, uint32_t part, uint32_t parts) { return feed->take_part(part, parts); }
bool bam_feed_next(BamFeed* feed, uint8_t* buffer, size_t capacity, agpu_bgzf_block* blocks, uint32_t block_capacity, ahost_bam_piece& piece) { return feed->next(buffer, capacity, blocks, block_capacity, piece); }

}
