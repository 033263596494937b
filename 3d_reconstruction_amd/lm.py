"""Levenberg-Marquardt on the same energy as fit_scene, with the points eliminated by a Schur
complement and the reduced camera system solved by a dense fp64 Cholesky on the GPU.

This is an EXTENSION: the reference optimises with first-order rules only (p2w:336-362).  It is
reached through the same entry point, `fit_scene(..., optimizer='LM')`, and returns the same list.
`lr` is the initial damping lambda.  One iteration:

    solve (J^T J + lambda diag(J^T J)) d = -J^T r      (Schur complement + Cholesky)
    trial = gauge_fix(params + d); evaluate E and J^T J there
    E decreased: accept, lambda /= 3      otherwise: restore, lambda *= 4 and try again

The trial point is written by the back-substitution into tensors of its own and evaluated into a second
set of gradient / block buffers, so an accepted step copies nothing and a rejected one neither copies nor
evaluates again: it goes back to the tensors and the evaluation it still has and rebuilds the camera
table (one small launch).

The gauge freedom (7 degrees) is handled the same way the reference's loop does it: the trial point
is mapped back to the gauge "camera 0 at the origin, longest baseline 1" (the energy is invariant),
and the damping keeps the normal equations positive definite.

Several GPUs: points are sharded like in the first-order path; the camera-side blocks, the
Schur complement and its right-hand side are summed over the ranks (engine.Shard.eval_jtj,
lm_step), every rank solves the reduced system itself and adopts the first rank's solution (the ranks'
own ones agree only to rounding), and the loop below runs identically on all ranks (the energy it
branches on is the cross-GPU sum)."""
from __future__ import annotations

import time

import numpy as np


class LMFit:
    def __init__(self, shard, max_iters, lam=1e-3, lam_down=1.0 / 3.0, lam_up=4.0, lam_min=1e-12, lam_max=1e12,
                 stop_value=0.0, stop_diff=1e-12, print_step=1000, printer=print):
        self.sh, self.max_iters = shard, int(max_iters)
        self.lam, self.lam_down, self.lam_up, self.lam_min, self.lam_max = lam, lam_down, lam_up, lam_min, lam_max
        self.stop_value, self.stop_diff, self.print_step, self.print = stop_value, stop_diff, print_step, printer
        self.E = np.zeros(self.max_iters)
        self.iters = 0
        self.accepted = self.rejected = 0
        self.stamps = []        # host clock after every energy read-back (which waits for the GPU): [start, E_0, E_1, ...]

    def energies(self):
        return self.E[:self.iters + 1].copy()

    def run(self):
        sh = self.sh
        self.stamps = [time.perf_counter()]
        sh.gauge_fix()
        sh.eval_jtj(post=True)
        E = sh.wait_energy()
        self.stamps.append(time.perf_counter())
        self.E[0] = E
        it = 0
        while it < self.max_iters - 1:
            if it % self.print_step == 0:
                self.print(f'{it} : {E}')
            if E < self.stop_value:
                self.print(f'{it} : Reached stopping value: {E} < {self.stop_value}')
                break
            if self.lam > self.lam_max:
                self.print(f'{it} : Reached maximum damping')
                break
            # One GPU: the step is queued without waiting for the factorisation's verdict, which is read after the
            # trial point's energy has been waited for anyway (a failed factorisation leaves a void trial point).
            queued = not sh.distributed
            if sh.lm_step(self.lam, trial=True, wait=not queued) != 0:          # not positive definite at this damping
                self.lam *= self.lam_up
                continue
            kept = sh.snapshot()
            sh.adopt(sh.X_trial, sh.cam_trial)
            sh.gauge_fix()                  # in place on the trial tensors
            sh.swap_evaluation()
            sh.eval_jtj(post=True)
            E_new = sh.wait_energy()        # (from the host mailbox: no copy, no wait for the stream)
            pivot = sh.lm_pivot() if queued else 0
            if pivot != 0:
                sh.restore(kept)
                sh.swap_evaluation()
                sh.rebuild_table()
                if pivot > 0:
                    self.lam *= self.lam_up
                continue
            self.stamps.append(time.perf_counter())
            it += 1
            if np.isfinite(E_new) and E_new < E:
                self.accepted += 1
                decrease = E - E_new
                E = E_new
                self.E[it] = E
                self.lam = max(self.lam * self.lam_down, self.lam_min)
                if 0.0 < decrease < self.stop_diff:
                    self.print(f'{it} : Reached stopping difference: {decrease} < {self.stop_diff}')
                    break
            else:
                self.rejected += 1
                sh.restore(kept)
                sh.swap_evaluation()
                sh.rebuild_table()
                self.E[it] = E
                self.lam *= self.lam_up
        else:
            self.print(f'{it} : Reached maximum iterations')
        self.iters = it
        self.print(f'E = {E}')
        return self
