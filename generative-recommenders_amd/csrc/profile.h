// Internal: event brackets for the benchmark's live kernel timing (see profile.hip).
#pragma once
#include <hip/hip_runtime.h>
bool hmx_profile_on();
void* hmx_profile_begin(const char* tag, hipStream_t s);
void hmx_profile_end(void* token, hipStream_t s);
