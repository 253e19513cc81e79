// TEST ONLY: round 3's forms of four places of the walk (csrc/bfs_device.h), kept so that tests/test_gpu_bfs_race.py can build the
// kernel that went wrong (1 walk in 20 000 beside a second context, every walk with -DMC_BFS_FUZZ) and require that it fails where the
// fixed kernel does not.  Included by bfs_device.h only under -DMC_BFS_OLD_RACE (metacherchant_amd/build.py variants fuzz_old,
// trace_old); the product library never sees this file.
#pragma once
// (1) no barrier between the waves' reads of plen / ppos / pdone -- which decide which barriers a wave meets -- and the stores to them
#define BFS_DECIDED_SYNC() do {} while (0)
// (2) the reset of those words in FRONT of the barrier: a wave late enough to read the new plen beside the old ppos gets
// avail = 2^32 - ppos, leaves the branch and is one barrier out of step with its workgroup from then on
#define BFS_PATH_RESET_EARLY(L, tid) do { if ((tid) < SCOUT_MAX_F) { (L).plen[tid] = 0; (L).ppos[tid] = 0; (L).pdone[tid] = 0; } } while (0)
#define BFS_PATH_RESET(L, tid) do {} while (0)
// (3), (4) the chunk counter read again in every iteration: behind the loop thread 0 resets it, and a wave that comes late to its
// last read sees 0 and starts over, alone
#define BFS_CHUNK_LOOP(c0) for (unsigned long long c0 = ctl_ld(&ctl->c0);; c0 = ctl_ld(&ctl->c0))
