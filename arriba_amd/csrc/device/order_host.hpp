// arriba_amd/csrc/device/order_host.hpp -- bucket counts of a std::unordered_map that receives n insertions one by one (host code).
// The counts come from the host's own std::__detail::_Prime_rehash_policy, i.e. from the very libstdc++ the reference is linked against.
#ifndef AGPU_ORDER_HOST_HPP
#define AGPU_ORDER_HOST_HPP 1

#include <stdint.h>
#include <unordered_map>
#include <utility>
#include <vector>

namespace agpu {

struct RehashPhase { uint32_t begin; uint64_t buckets; }; // rehash to `buckets` right before insertion number `begin`

inline std::vector<RehashPhase> rehash_schedule(uint32_t n) {
	std::vector<RehashPhase> phases;
	std::__detail::_Prime_rehash_policy policy; // max_load_factor 1.0, as in the reference's default-constructed map
	std::size_t buckets = 1;                    // a default-constructed unordered_map has a single bucket
	std::size_t elements = 0;
	while (elements < n) {
		std::pair<bool, std::size_t> rehash = policy._M_need_rehash(buckets, elements, 1);
		if (rehash.first) {
			buckets = rehash.second;
			RehashPhase phase; phase.begin = (uint32_t) elements; phase.buckets = buckets;
			phases.push_back(phase);
		}
		// nothing can change before the element count exceeds the policy's next threshold
		std::size_t next = policy._M_next_resize;
		elements = (next > elements) ? next : elements + 1;
	}
	return phases;
}

}

#endif
