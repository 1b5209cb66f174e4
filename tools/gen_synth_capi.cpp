// tools/gen_synth_capi.cpp -- C entry points of the synthetic generator for Python (tests, bench.py).  Test/bench tooling.
#include <cstring>
#include <string>
#include <vector>
#include "gen_synth.hpp"

extern "C" {

struct synth_handle { synth::Generator* generator; std::vector<uint8_t> bam; std::string error; };

// options: "key=value key=value ..." with the Config field names used by the gen_synth command line
synth_handle* synth_create(unsigned long long seed, long fragments, double normal_multiplier, int contigs, int contig_length, double genes_per_mb, int junctions,
                           int read_length, int clip_min, int clip_max, double partner_clip, int shuffle, int separate_mates, int stranded) {
	synth::Config config;
	config.seed = seed; config.fragments = fragments; config.normal_multiplier = normal_multiplier; config.contigs = contigs; config.contig_length = contig_length;
	config.genes_per_mb = genes_per_mb; config.junctions = junctions; config.read_length = read_length; config.clip_min = clip_min; config.clip_max = clip_max;
	config.frac_clip_from_partner = partner_clip; config.shuffle_names = shuffle != 0; config.separate_mates = separate_mates != 0; config.stranded = stranded != 0;
	synth_handle* handle = new synth_handle();
	try {
		handle->generator = new synth::Generator(config);
		handle->generator->build_reference();
	} catch (const std::exception& e) {
		handle->error = e.what();
		handle->generator = NULL;
	}
	return handle;
}
const char* synth_error(synth_handle* handle) { return handle->error.c_str(); }
int synth_write_reference(synth_handle* handle, const char* fasta_path, const char* gtf_path) {
	try { handle->generator->write_fasta(fasta_path); handle->generator->write_gtf(gtf_path); return 0; }
	catch (const std::exception& e) { handle->error = e.what(); return -1; }
}
int synth_write_bam(synth_handle* handle, const char* bam_path) {
	try { handle->generator->write_bam(bam_path); return 0; }
	catch (const std::exception& e) { handle->error = e.what(); return -1; }
}
// generates the raw (inflated) BAM stream in memory; returns its size
unsigned long long synth_generate_bam(synth_handle* handle) {
	handle->bam.clear();
	handle->generator->stream_bam([&](const uint8_t* data, size_t size) { handle->bam.insert(handle->bam.end(), data, data + size); });
	return handle->bam.size();
}
const uint8_t* synth_bam_data(synth_handle* handle) { return handle->bam.data(); }
long synth_records(synth_handle* handle) { return handle->generator->records_written(); }
void synth_destroy(synth_handle* handle) { if (handle) { delete handle->generator; delete handle; } }

}
