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
#elif AHA_UNIT_LAB == 8   // expansion: 512 threads x 2 records
#define AHA_LAB_XG_SHAPE 512, 2, 1536
#elif AHA_UNIT_LAB == 9   // expansion: 1024 threads x 1 record
#define AHA_LAB_XG_SHAPE 1024, 1, 1536
#elif AHA_UNIT_LAB == 10  // expansion: 256 threads x 8 records, 3072 staged hits (most groups in one block)
#define AHA_LAB_XG_SHAPE 256, 8, 3072
#elif AHA_UNIT_LAB == 11  // expansion: hits staged but not stored
#define AHA_LAB_XG_NO_STORE 1
#elif AHA_UNIT_LAB == 12  // expansion: no gather of the END state's key
#define AHA_LAB_XG_NO_GATHER 1
#elif AHA_UNIT_LAB == 13  // expansion: 128 threads x 4 records, 768 staged hits (more workgroups per CU)
#define AHA_LAB_XG_SHAPE 128, 4, 768
#elif AHA_UNIT_LAB == 15  // expansion: 1024 threads x 2 records, 3072 staged hits
#define AHA_LAB_XG_SHAPE 1024, 2, 3072
#elif AHA_UNIT_LAB == 17  // expansion: 512 threads x 1 record, 768 staged hits
#define AHA_LAB_XG_SHAPE 512, 1, 768
#elif AHA_UNIT_LAB == 18  // expansion: 1024 threads x 1 record, 3072 staged hits
#define AHA_LAB_XG_SHAPE 1024, 1, 3072
#elif AHA_UNIT_LAB == 19  // expansion: 1024 x 1, no gather
#define AHA_LAB_XG_SHAPE 1024, 1, 1536
#define AHA_LAB_XG_NO_GATHER 1
#elif AHA_UNIT_LAB == 20  // expansion: 1024 x 1, no store
#define AHA_LAB_XG_SHAPE 1024, 1, 1536
#define AHA_LAB_XG_NO_STORE 1
#elif AHA_UNIT_LAB == 21  // expansion 1024 x 1: records loaded and hits stored non-temporally
#define AHA_LAB_XG_SHAPE 1024, 1, 1536
#define AHA_LAB_NT_REC 1
#define AHA_LAB_NT_HIT 1
#elif AHA_UNIT_LAB == 22  // ... hits only
#define AHA_LAB_XG_SHAPE 1024, 1, 1536
#define AHA_LAB_NT_HIT 1
#elif AHA_UNIT_LAB == 23  // ... records only
#define AHA_LAB_XG_SHAPE 1024, 1, 1536
#define AHA_LAB_NT_REC 1
#elif AHA_UNIT_LAB == 24  // traversal: the text loaded non-temporally
#define AHA_LAB_NT_TEXT 1
#elif AHA_UNIT_LAB == 25  // traversal: the event buffers stored non-temporally
#define AHA_LAB_NT_EV 1
#elif AHA_UNIT_LAB == 26  // traversal: both
#define AHA_LAB_NT_TEXT 1
#define AHA_LAB_NT_EV 1
#elif AHA_UNIT_LAB == 27  // expansion 1024 x 1: gathers folded into 8 KiB
#define AHA_LAB_XG_SHAPE 1024, 1, 1536
#define AHA_LAB_XG_GATHER_INDEX(i) ((i) & 0x3FFu)
#elif AHA_UNIT_LAB == 28  // ... into 512 KiB
#define AHA_LAB_XG_SHAPE 1024, 1, 1536
#define AHA_LAB_XG_GATHER_INDEX(i) ((i) & 0xFFFFu)
#elif AHA_UNIT_LAB == 30  // expansion: no LDS cache of uend entries
#define AHA_LAB_XG_NO_CACHE 1
#elif AHA_UNIT_LAB == 31  // expansion: records loaded non-temporally
#define AHA_LAB_NT_REC 1
#elif AHA_UNIT_LAB == 32  // expansion: one workgroup per group instead of persistent ones
#define AHA_LAB_XG_GRID 65536
#elif AHA_UNIT_LAB == 33  // expansion: 256 x 4 with the cache, 8 workgroups per CU
#define AHA_LAB_XG_SHAPE 256, 4, 1536
#define AHA_LAB_XG_GRID 2048
#elif AHA_UNIT_LAB == 34  // expansion: 512 x 2 with the cache, 4 workgroups per CU
#define AHA_LAB_XG_SHAPE 512, 2, 1536
#define AHA_LAB_XG_GRID 1024
#elif AHA_UNIT_LAB == 14  // expansion: 512 threads x 4 records, 3072 staged hits
#define AHA_LAB_XG_SHAPE 512, 4, 3072
#endif
