// cedar_replay.hpp -- stale END flags of the reference's Cedar (src/aha/cedar.cr:642-648), see cedar_replay.cpp.
#pragma once
#include <cstdint>
#include <vector>

#include "automaton.hpp"

namespace aha {

// States of `a` (ids, ascending) whose node in the reference's double array -- built by inserting a's keys in
// compile order -- answers is_end? (cedar.cr:657-660) with true although it holds no key.  Read by match_longest only.
void cedar_stale_ends(const Automaton &a, std::vector<uint32_t> &stale_states);

}  // namespace aha
