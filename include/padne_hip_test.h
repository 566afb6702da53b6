/* padne_hip_test.h -- TEST-ONLY entry points of libpadne_hip.so.  Not part of the drop-in boundary (include/padne_hip.h):
 * nothing a padne integration binds lives here.  The library exports them so that the row-partitioned solver
 * (halo plans, one multigrid hierarchy over all ranks, the collectives inside the CG loop) can be driven with several
 * ranks on a ONE-GPU box, where RCCL refuses two ranks on the same device. */
#ifndef PADNE_HIP_TEST_H
#define PADNE_HIP_TEST_H

#include "padne_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* In-process team: several contexts of ONE process on ONE GPU act as ranks (one host thread per context drives
 * its solve); all-reduce / all-gather go through host barriers and peer copies instead of RCCL, which refuses two
 * ranks on one device.  For rehearsing the row-partitioned solver on a single-GPU box (tests); sums are formed in
 * rank order, so every rank sees the same bits, like with RCCL. */
int padne_team_create(int world_size, void **team_out);
int padne_team_destroy(void *team);
int padne_ctx_join_team(padne_ctx *ctx, void *team, int rank);
/* Marks the team failed and wakes every rank that waits in a collective: all pending and future team collectives
 * return PADNE_E_COMM.  Called by the driver of a rank that leaves early (error, exception), so that its peers do not
 * wait for it for ever. */
int padne_team_abort(void *team);
/* Interior / boundary tiles of a row-partitioned operator once a solve has examined it (64-row tiles whose columns are
 * all owned / that read an exchange slot); both 0 while there is no split plan.  level < 0: the matrix itself, otherwise
 * the operator A_level of its cached hierarchy. */
int padne_csr_split_tiles(const padne_csr *m, int level, int64_t *interior, int64_t *boundary);
/* Groups of right-hand sides (eight, or five to seven zero-padded) this context has advanced in lockstep through the
 * batched multigrid-PCG so far -- what tells a test that the K + 1 right-hand sides of K regulators took that path. */
int padne_ctx_lockstep_groups(const padne_ctx *ctx, int64_t *groups);
/* Assemblies of this process whose rows were built by the two-pass second path of the row kernel (count, scan, fill):
 * forced with PADNE_FORCE=asm_two_pass, or taken after the single-pass kernel's in-kernel scan gave up on a shared chip. */
int padne_asm_second_path_count(int64_t *count);
/* Reads the PADNE_* environment switches again for a context that lives across a change of them (the library reads them
 * once, when a context is created): the tests flip a switch, reload, and compare the two paths on one context. */
int padne_ctx_reload_options(padne_ctx *ctx);
/* Average device time of one halo exchange of the context's plan (padne_ctx_set_halo) over `repeats` exchanges queued back to
 * back -- peer-to-peer stores into the shared mailboxes with their device-side flags, or the all-gather.  Collective: every
 * rank calls it with the same count (tests/two_process_rank.py, bench.py rank_proxy.p2p_exchange). */
int padne_ctx_halo_exchange_time(padne_ctx *ctx, int32_t repeats, double *seconds_out);

#ifdef __cplusplus
}
#endif

#endif /* PADNE_HIP_TEST_H */
