/*
 * oracle/arena.cpp -- TEST INFRASTRUCTURE, not product code.
 *
 * Parity hazard H1: the reference keeps gene/exon sets as vectors of raw
 * pointers sorted BY POINTER VALUE (source/common.hpp:128-146,156-160), so genes[0],
 * set_intersection and the gene1 x gene2 loop order of find_fusions follow heap addresses of
 * std::list nodes.  To make that order well defined the oracle build replaces the global
 * operator new: allocations whose size equals the list-node size of a gene, exon or
 * transcript annotation record are served from a never-reusing bump arena, so pointer order ==
 * allocation order == GTF order for real genes/exons and creation order for dummy genes
 * (source/arriba.cpp:249,321-325).  Everything else goes to malloc/free unchanged.
 * ARRIBA_ORACLE_ARENA=0 disables the arena (to diff against plain glibc malloc order).
 */
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <new>
#include <list>
#include <sys/mman.h>
#include "common.hpp"

namespace {

const size_t ARENA_BYTES = (size_t) 32 << 30; // virtual reservation; pages are touched lazily
char* arena_base = NULL;
char* arena_next = NULL;
char* arena_end = NULL;
int arena_state = -1; // -1 = not initialised, 0 = disabled, 1 = enabled

const size_t NODE_SIZE_GENE = sizeof(std::_List_node<gene_annotation_record_t>);
const size_t NODE_SIZE_EXON = sizeof(std::_List_node<exon_annotation_record_t>);
const size_t NODE_SIZE_TRANSCRIPT = sizeof(std::_List_node<transcript_annotation_record_t>);

void arena_init() {
	const char* setting = getenv("ARRIBA_ORACLE_ARENA");
	if (setting != NULL && setting[0] == '0') {
		arena_state = 0;
		return;
	}
	void* p = mmap(NULL, ARENA_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
	if (p == MAP_FAILED) {
		arena_state = 0;
		return;
	}
	arena_base = arena_next = (char*) p;
	arena_end = arena_base + ARENA_BYTES;
	arena_state = 1;
}

inline void* allocate(size_t size) {
	if (arena_state < 0)
		arena_init();
	if (arena_state == 1 && (size == NODE_SIZE_GENE || size == NODE_SIZE_EXON || size == NODE_SIZE_TRANSCRIPT)) {
		size_t rounded = (size + 15) & ~(size_t) 15;
		if (arena_next + rounded <= arena_end) {
			void* p = arena_next;
			arena_next += rounded;
			return p;
		}
	}
	void* p = malloc(size ? size : 1);
	if (p == NULL)
		throw std::bad_alloc();
	return p;
}

inline void release(void* p) {
	if (p == NULL)
		return;
	if ((char*) p >= arena_base && (char*) p < arena_end)
		return; // arena memory is never reused
	free(p);
}

}

void* operator new(size_t size) { return allocate(size); }
void* operator new[](size_t size) { return allocate(size); }
void* operator new(size_t size, const std::nothrow_t&) noexcept { try { return allocate(size); } catch (...) { return NULL; } }
void* operator new[](size_t size, const std::nothrow_t&) noexcept { try { return allocate(size); } catch (...) { return NULL; } }
void operator delete(void* p) noexcept { release(p); }
void operator delete[](void* p) noexcept { release(p); }
void operator delete(void* p, const std::nothrow_t&) noexcept { release(p); }
void operator delete[](void* p, const std::nothrow_t&) noexcept { release(p); }
void operator delete(void* p, size_t) noexcept { release(p); }
void operator delete[](void* p, size_t) noexcept { release(p); }
