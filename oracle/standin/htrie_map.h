/*
 * oracle/standin/htrie_map.h -- TEST INFRASTRUCTURE, not product code.
 * Stand-in for Tessil hat-trie v0.6.0 (reference Makefile:22-24), which the
 * reference uses only as a name -> bam1_t* collation map
 * (source/read_chimeric_alignments.cpp:16,605,677-680). Surface needed:
 *   insert(const char*, T) -> pair<iterator,bool>, erase(iterator), *iterator -> T&.
 * Backed by std::unordered_map; contributes no arithmetic to the results.
 */
#ifndef ORACLE_STANDIN_HTRIE_MAP_H
#define ORACLE_STANDIN_HTRIE_MAP_H 1
#include <string>
#include <unordered_map>
#include <utility>
namespace tsl {
template <class CharT, class T> class htrie_map {
	typedef std::unordered_map<std::basic_string<CharT>, T> base_t;
	base_t m;
public:
	class iterator {
		friend class htrie_map;
		typename base_t::iterator it;
	public:
		iterator() {}
		explicit iterator(typename base_t::iterator i): it(i) {}
		T& operator*() const { return it->second; }
		T* operator->() const { return &it->second; }
		bool operator==(const iterator& o) const { return it == o.it; }
		bool operator!=(const iterator& o) const { return it != o.it; }
	};
	std::pair<iterator,bool> insert(const CharT* key, const T& value) {
		std::pair<typename base_t::iterator,bool> r = m.emplace(key, value);
		return std::make_pair(iterator(r.first), r.second);
	}
	void erase(iterator pos) { m.erase(pos.it); }
	size_t size() const { return m.size(); }
	bool empty() const { return m.empty(); }
};
}
#endif
