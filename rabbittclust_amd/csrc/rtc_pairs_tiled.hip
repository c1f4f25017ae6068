// rtc_pairs_tiled.hip -- LDS mask-table tiles (placeholder until the tiled kernel lands).
#include "rtc_internal.h"
int rtc_pair_common_tiled(rtc_ctx*, const void*, int, const uint64_t*, const uint32_t*, uint32_t, uint32_t,
                          uint32_t, uint32_t, uint32_t, uint32_t*, uint64_t, int, int* handled) {
  *handled = 0;
  return RTC_OK;
}
