/* gpv_testhooks.h -- fault injection for the tests of the fail-closed verdict. NOT part of the public boundary (include/gpv.h) and NOT
 * in the product library: only libgpv_test.so (the same objects, gpv_api.cpp recompiled with -DGPV_TEST_HOOKS; csrc/Makefile) defines
 * and exports it. Process-wide; disarm with stage = 0. */
#pragma once
#pragma GCC visibility push(default)
#ifdef __cplusplus
extern "C" {
#endif
/* The `nth` launch (counted from arming; -1 = every launch) of pipeline stage `stage` (GPV_STAGE_* of gpv_launch.h: 1 range check,
 * 2 transcript, 3 plonk, 4 FRI queries, 5 Merkle leaves, 6 sibling walk, 7 crown plan, 8 crown reconcile, 9 crown level, 10 crown
 * finish, 11 derive_extra) keeps only num / den of its grid; 0 / 1 skips the launch. stage 100: rank `nth` of a gpv_group reports a
 * failed verification of its block (the other ranks must return GPV_EPEER instead of waiting for it). */
int gpvi_test_set_fault(int stage, int nth, unsigned num, unsigned den);
#ifdef __cplusplus
}
#endif
#pragma GCC visibility pop
