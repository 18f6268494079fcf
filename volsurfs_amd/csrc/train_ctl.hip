// Device-side control of a graph-replayed training iteration (include/volsurfs_hip.h: vsa_train_ctl).
//
// What /root/reference/volsurfs_py/trainer.py does on the host between two iterations — the dynamic ray count
// (:288-304: nr_rays = int(nr_rays * (target / nr_samples))), the scheduler step (:306-308: linear warm-up
// schedulers/warmup.py:26-47 with multiplier 1, then MultiStepLR(gamma 0.3), base_method.py:71-76), Adam's step count
// (base_method.py:87-94) — and what the sampler's random stream does (TensorReel.get_next_rays_batch, :176-190: advanced
// by 2^32 per call, as src/RaySampler.cu:139-142) as ONE lane at the end of the iteration, in the same double / integer
// arithmetic as volsurfs_amd/trainer.py::dynamic_nr_rays and volsurfs_amd/schedulers.py::lr_at, so that the graph loop
// and the eager loop draw the same batch sizes and learning rates.
#include "common.h"
#include "pcg32.h"

namespace {

__device__ float ctl_lr_at(const vsa_train_ctl& c, int it) {
  int decay_epoch = it;
  if (c.nr_warmup > 0) {
    if (it <= c.nr_warmup) return (float)(c.lr_base * ((double)it / (double)c.nr_warmup));
    decay_epoch = it - c.nr_warmup - 1;
    if (decay_epoch < 0) decay_epoch = 0;
  }
  int k = 0;                                  // bisect_right(milestones, decay_epoch)
  while (k < c.nr_milestones && c.milestone[k] <= decay_epoch) ++k;
  return c.lr_stage[k];
}

__global__ void train_ctl_tick_kernel(vsa_train_ctl* ctl) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  vsa_train_ctl c = *ctl;
  // the update of the iteration that has just finished is now pending: its lr is the one the schedule holds
  // BEFORE this iteration's scheduler step (trainer.py steps the scheduler after the optimiser)
  c.adam_step += 1;
  c.adam_pending = 1;
  c.adam_lr = ctl_lr_at(c, c.iter);
  c.sum_rays += c.nr_rays;
  c.sum_hits += c.nr_hits;
  // dynamic ray count
  int n = c.nr_rays;
  if (c.target_hits > 0 && c.nr_hits > 0)
    n = (int)((double)n * ((double)c.target_hits / (double)c.nr_hits));
  if (n > c.capacity) {
    n = c.capacity;
    c.clamped += 1;
  }
  if (n < 1) n = 1;
  c.nr_rays = n;
  c.loss_scale = (float)(c.loss_weight / (3.0 * (double)n));
  c.iter += 1;
  Pcg32 rng{c.rng_state, c.rng_inc};
  rng.advance(1ull << 32);
  c.rng_state = rng.state;
  *ctl = c;
}

}  // namespace

extern "C" int vsa_train_ctl_tick(vsa_train_ctl* ctl, void* stream) {
  if (!ctl) return VSA_ERR_ARG;
  hipLaunchKernelGGL(train_ctl_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctl);
  VSA_RETURN_LAUNCH_STATUS();
}
