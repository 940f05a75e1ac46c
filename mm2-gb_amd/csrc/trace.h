// trace.h -- internal: named ranges around the host-side stages of a micro-batch, visible in rocprofv3 --marker-trace
// (role of the roctxRangePush/Pop pairs in the reference, gpu/plchain.cu:349-412).  The marker library is looked up at run time
// (librocprofiler-sdk-roctx.so, then libroctx64.so); without it, or with MM2GB_NO_ROCTX set, a range costs one branch.
#pragma once
#include <dlfcn.h>
#include <cstdlib>

namespace mm2gb {

struct TraceApi {
	int (*push)(const char*) = nullptr;
	int (*pop)() = nullptr;
	TraceApi()
	{
		if (getenv("MM2GB_NO_ROCTX")) return;
		for (const char *name : { "librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4" }) {
			void *h = dlopen(name, RTLD_LAZY | RTLD_LOCAL);
			if (!h) continue;
			push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
			pop = (int (*)())dlsym(h, "roctxRangePop");
			if (push && pop) return;
			push = nullptr; pop = nullptr;
		}
	}
};

inline const TraceApi &trace_api() { static const TraceApi api; return api; }

struct TraceRange {
	bool on;
	explicit TraceRange(const char *name) : on(trace_api().push != nullptr) { if (on) trace_api().push(name); }
	~TraceRange() { if (on) trace_api().pop(); }
	TraceRange(const TraceRange&) = delete;
	TraceRange &operator=(const TraceRange&) = delete;
};

} // namespace mm2gb
