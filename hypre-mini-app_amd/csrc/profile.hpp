// HIP-event timing of selected kernel classes on the library's own stream
// (bench.py's roofline.achieved comes from these; torch.cuda.Event would only
// see torch's current stream).
#pragma once
#include "kernels.hpp"

namespace mi {

struct KernelTimer {
  bool enabled[k::PROF_COUNT] = {};
  std::vector<hipEvent_t> start[k::PROF_COUNT], stop[k::PROF_COUNT];
  size_t used[k::PROF_COUNT] = {};
  size_t dropped[k::PROF_COUNT] = {};
  // instantiation (template flags included) of the kernel last launched under the class, as rocprofv3 names it
  const char *kernel_name[k::PROF_COUNT] = {};
  void enable(int id, size_t capacity);
  void reset();
  // (launch count, total milliseconds) of completed pairs; synchronises the stream
  void collect(int id, long long *count, double *total_ms, double *min_ms);
  ~KernelTimer();
};

inline void prof_begin(int id, hipStream_t s) {
  if (id < 0) return;
  KernelTimer *t = ctx().timer;
  if (!t || !t->enabled[id]) return;
  if (t->used[id] >= t->start[id].size()) {
    t->dropped[id]++;
    return;
  }
  (void)hipEventRecord(t->start[id][t->used[id]], s);
}
inline void prof_name(int id, const char *name) {
  if (id < 0) return;
  KernelTimer *t = ctx().timer;
  if (t && t->enabled[id]) t->kernel_name[id] = name;
}
inline void prof_end(int id, hipStream_t s) {
  if (id < 0) return;
  KernelTimer *t = ctx().timer;
  if (!t || !t->enabled[id]) return;
  if (t->used[id] >= t->start[id].size()) return;
  (void)hipEventRecord(t->stop[id][t->used[id]], s);
  t->used[id]++;
}

}  // namespace mi
