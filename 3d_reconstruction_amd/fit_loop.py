"""Host-side control of the optimisation loop: learning-rate schedule with snapshot / rollback,
optimiser state handling, stopping rules and logging.  Behaviour follows the reference's
`fit_scene` loop (p2w:592-726) and `lr_manager` (p2w:365-454) decision for decision; all array
work is delegated to a shard object (engine.Shard) whose buffers live on the GPU.

The only device-to-host traffic per iteration is the 8-byte energy, which the schedule
branches on."""
from __future__ import annotations

import numpy as np

RHO_1, RHO_2 = 0.9, 0.999      # p2w:351-352

# optimiser registry, looked up by name like the reference's gd_func_dict (p2w:362, 527, 705);
# the values name the device update rule the shard runs.
gd_func_dict = {'SGD': 'ba_sgd_step', 'Adam': 'ba_adam_step'}

# optimisers beyond the reference's registry, accepted by fit_scene(optimizer=...)
extra_optimizers = {'LM': 'Levenberg-Marquardt with Schur complement (lm.LMFit)'}


class SceneFit:
    """One `fit_scene` run as an object: `step()` executes one pass of the loop body and
    returns False when the loop is over."""

    def __init__(self, shard, lr, max_iters, X_mask=True, C_Theta_mask=True, phi_x_mask=True,
                 optimizer='Adam', patience=1000, factor=2.0, stop_value=0.0, stop_diff=1e-12,
                 print_step=1000, ret_all_iters=False, printer=print):
        if optimizer not in gd_func_dict:
            raise ValueError(f'Unkown name of the optmizer, choose one from {[key for key in gd_func_dict]}')
        self.sh = shard
        self.lr, self.max_iters = lr, int(max_iters)
        self.masks = (bool(X_mask), bool(C_Theta_mask), bool(phi_x_mask))
        self.optimizer = optimizer
        self.patience, self.factor = patience, factor
        self.stop_value, self.stop_diff, self.print_step = stop_value, stop_diff, print_step
        self.ret_all_iters = ret_all_iters
        self.print = printer
        self.timed_every = 0        # record HIP events around the observation-pass kernel of every n-th pass
        self._passes = 0
        self.gauge = self.masks[0] and self.masks[1]      # p2w:533, 710

        K6 = 6 * shard.K
        lo = 0 if self.masks[1] else K6
        hi = (K6 + 1) if self.masks[2] else K6
        self.cam_range = (lo, hi) if hi > lo else None     # free slice of [C | Theta | phi_x]

        if self.gauge:
            shard.gauge_fix()                                            # p2w:533-536

        # free parameter groups as (parameters, gradient) tensor pairs
        self.groups = []
        if self.masks[0]:
            self.groups.append((shard.X.view(-1), shard.gX.view(-1)))
        if self.cam_range:
            self.groups.append((shard.cam[lo:hi], shard.red[lo:hi]))
        self.release = [p.clone() for p, _ in self.groups]              # p2w:463-464
        self.adam = optimizer == 'Adam'
        if self.adam:                                                    # p2w:469-473
            self.s = [p.new_zeros(p.shape) for p, _ in self.groups]
            self.r = [p.new_zeros(p.shape) for p, _ in self.groups]
            self.t = 1
            self.s_tmp = [p.new_zeros(p.shape) for p, _ in self.groups]
            self.r_tmp = [p.new_zeros(p.shape) for p, _ in self.groups]
            self.t_tmp = 1
        self.all_iters = self.max_iters * [None] if ret_all_iters else None

        self.E = np.zeros(self.max_iters)                                # p2w:592-599
        self.E_min = self.E_min_temp = np.inf
        self.patience_timer, self.success_timer = 0, patience
        self.began_decreasing = self.increased = False
        self.iters = 0
        self.finished = False
        self._snapshot = False      # a snapshot of the current parameters is due with the next update

    # ---- optimiser state moves (p2w:411-414, 439-446) -----------------------------------------
    def _adam_reset(self):
        if self.adam:
            for s, r in zip(self.s, self.r):
                s.zero_()
                r.zero_()
            self.t = 1

    def _adam_stash_and_reset(self):
        if self.adam:
            for k in range(len(self.groups)):
                self.s_tmp[k].copy_(self.s[k])
                self.r_tmp[k].copy_(self.r[k])
            self.t_tmp = self.t
            self._adam_reset()

    def _adam_restore(self):
        if self.adam:
            for k in range(len(self.groups)):
                self.s[k].copy_(self.s_tmp[k])
                self.r[k].copy_(self.r_tmp[k])
            self.t = self.t_tmp

    def _evaluate(self):
        timed = self.timed_every > 0 and self._passes % self.timed_every == 0
        self._passes += 1
        self.sh.eval(grad_x=self.masks[0], grad_cam=self.masks[1], grad_phi=self.masks[2], timed=timed)

    # ---- learning-rate schedule ---------------------------------------------------------------
    def _manage(self):
        """The reference's lr_manager for the energy just stored in E[iters]."""
        entry = self.iters
        e = self.E[entry]
        if entry == 0:
            self.E_min = e
        elif e < self.E_min:
            self.E_min = e
            self.began_decreasing = True
            # release <- main (p2w:397-398); the copy is written by the update kernels, which read
            # the old parameters anyway.  Nothing reads `release` before the next call of _manage.
            self._snapshot = True
            self.patience_timer = self.patience
            if self.success_timer > 0:
                self.success_timer -= 1
            else:
                self.print(f'{self.iters} : Icrease LR to {self.lr * self.factor}')
                self.lr *= self.factor
                self.success_timer = self.patience
                self.increased = True
                self.E_min_temp = self.E_min
                self._adam_stash_and_reset()
        elif e > self.E_min:
            if self.increased:
                self.patience_timer = 0
                self.iters -= 1
            self.success_timer = self.patience
            if self.patience_timer > 0:
                self.patience_timer -= 1
            else:
                if self.began_decreasing:
                    self.patience_timer = self.patience
                else:
                    self.iters -= 1
                self.print(f'{self.iters} : Decrease LR to {self.lr / self.factor}')
                self.lr /= self.factor
                for rel, (p, _) in zip(self.release, self.groups):
                    p.copy_(rel)
                self.sh.invalidate_table()
                self._evaluate()          # gradients at the rolled-back parameters
                if (not self.increased) or (self.E_min_temp > self.E_min):
                    self._adam_reset()
                else:
                    self._adam_restore()
                    self.increased = False
        if self.ret_all_iters:
            self.all_iters[entry] = [p.clone() for p, _ in self.groups]

    # ---- one pass of the loop body (p2w:600-714) ----------------------------------------------
    def step(self):
        if self.finished or not (self.iters < self.max_iters):
            self.finished = True
            return False
        self._evaluate()
        self.E[self.iters] = self.sh.energy()
        self._manage()

        it = self.iters
        if self.lr == 0.0:
            self.print(f'{it} : Reached lr = 0.0')
            return self._finish()
        if self.stop_diff > 0.0 and it > 2 * self.patience:
            diff = self.E[:it - 2 * self.patience].min() - self.E_min
            if 0.0 < diff < self.stop_diff:
                self.print(f'{it} : Reached stopping difference: {diff} < {self.stop_diff}')
                return self._finish()
        if self.E[it] < self.stop_value:
            self.print(f'{it} : Reached stopping value: {self.E[it]} < {self.stop_value}')
            return self._finish()
        if it == self.max_iters - 1:
            self.print(f'{it} : Reached maximum iterations')
            return self._finish()
        if it % self.print_step == 0:
            self.print(f'{it} : {self.E_min}')

        self.update()
        self.iters += 1
        return True

    def update(self):
        """Update rule on every free group, pending snapshot, gauge fix (p2w:704-712)."""
        bc1 = bc2 = 1.0
        if self.adam:
            bc1, bc2 = 1.0 - RHO_1 ** self.t, 1.0 - RHO_2 ** self.t
        gx = 0 if self.masks[0] else None                 # index of the X group
        gc = (1 if self.masks[0] else 0) if self.cam_range else None
        mom = lambda k: (self.s[k], self.r[k]) if (self.adam and k is not None) else None
        rel = lambda k: self.release[k] if (self._snapshot and k is not None) else None
        self.sh.step_update(2 if self.adam else 1, self.lr, bc1, bc2, self.masks[0], self.cam_range,
                            mom(gx), mom(gc), rel(gx), rel(gc), self.gauge)
        self._snapshot = False
        if self.adam:
            self.t += 1

    def _finish(self):
        self.finished = True
        return False

    def run(self):
        while self.step():
            pass
        self.print(f'E = {self.E_min}')                                   # p2w:716
        return self

    # ---- results ------------------------------------------------------------------------------
    def energies(self):
        return self.E[:self.iters + 1].copy()                             # p2w:723

    def history(self):
        """Per-iteration parameter copies as numpy arrays, one list per free tensor group
        (X; then the free slice of [C | Theta | phi_x])."""
        if not self.ret_all_iters:
            return None
        return [[None if snap is None else snap[k].cpu().numpy() for snap in self.all_iters]
                for k in range(len(self.groups))]
