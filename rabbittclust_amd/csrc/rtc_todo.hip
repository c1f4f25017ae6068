// temporary: entry points declared in include/rtclust.h that are not implemented yet
#include "rtc_internal.h"
extern "C" {
int rtc_greedy(rtc_ctx* ctx, const void*, int, const uint64_t*, const uint32_t*, uint32_t, const uint32_t*, int, int, int, double, int32_t*, uint32_t*) { return rtc_fail(ctx, RTC_ERR_UNSUPPORTED, "not implemented"); }
}
