// Lab variants of ku_traverse (scan_unit.hip), compiled into libaha_hip_lab<k>.so by `make -C aha_amd/csrc lab K=<k>` with
// -DAHA_LAB_INCLUDE='"../../tools/lab/unit_lab.hpp"' -DAHA_UNIT_LAB=<k>.  Timing only: they break the walk on purpose, the
// product library never includes this file.
#pragma once
#if AHA_UNIT_LAB == 1    // every probe lands in the first 64 Ki slots (512 KiB: L2 hits)
#define AHA_LAB_PROBE_INDEX(i) ((i) & 0xFFFFu)
#elif AHA_UNIT_LAB == 2  // ... in the first 1 Ki slots (8 KiB: L1 hits)
#define AHA_LAB_PROBE_INDEX(i) ((i) & 0x3FFu)
#elif AHA_UNIT_LAB == 3  // no probe at all: the floor of the trip
#define AHA_LAB_NO_PROBE 1
#elif AHA_UNIT_LAB == 4  // nothing is reported: the cost of the event path
#define AHA_LAB_NO_EVENTS 1
#elif AHA_UNIT_LAB == 5  // two-walk kernel: the 16 input bytes are loaded (and waited for) by the refill that needs them
#define AHA_LAB_W2_BLOCKQ 1
#elif AHA_UNIT_LAB == 6  // no probe, nothing reported: the walk's instruction floor
#define AHA_LAB_NO_PROBE 1
#define AHA_LAB_NO_EVENTS 1
#elif AHA_UNIT_LAB == 7  // a full event buffer is stored where it fills (round 3's way), not behind the next trip's probe
#define AHA_LAB_IMMEDIATE_FLUSH 1
#endif
