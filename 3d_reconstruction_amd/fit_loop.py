"""Host-side control of the optimisation loop: learning-rate schedule with snapshot / rollback,
optimiser state handling, stopping rules and logging.  Behaviour follows the reference's
`fit_scene` loop (p2w:592-726) and `lr_manager` (p2w:365-454) decision for decision; all array
work is delegated to a shard object (engine.Shard) whose buffers live on the GPU.

The only device-to-host traffic per iteration is the 8-byte energy, which the schedule
branches on.  To keep the GPU busy while that value travels, the update that follows an
evaluation is queued BEFORE the energy is read, with the learning rate and moments the schedule
will use unless it changes them (LR increase, LR decrease with rollback, a stopping rule).  The
step writes new buffers and leaves its inputs alone, so when the schedule does intervene the
queued step is simply dropped and the loop continues from the untouched state: the sequence of
parameters is the one the reference computes either way.

When the shard's observation pass can apply the point update itself (`can_fuse`), the queued step
is: pass with the X update fused in, then the camera-side step; the rigid part of the gauge fix
stays pending on the new points (a 16-double record) and is applied by the next pass while it
loads them."""
from __future__ import annotations

import numpy as np

RHO_1, RHO_2 = 0.9, 0.999      # p2w:351-352



def sgd(x, grad_x, lr):
    """Plain gradient step on host arrays, the reference's `sgd` (p2w:336-342): (x_new, {})."""
    return x - lr * grad_x, {}


def gd_adam(x, grad_x, lr, s, r, t, rho_1=RHO_1, rho_2=RHO_2):
    """Adam on host arrays with the reference's arithmetic (p2w:344-360): bias-corrected moments, the
    1e-12 added OUTSIDE the square root, `t` a one-element integer array that counts from 1.
    Returns (x_new, {'s', 'r', 't'})."""
    s_new = rho_1 * s + (1.0 - rho_1) * grad_x
    r_new = rho_2 * r + (1.0 - rho_2) * grad_x ** 2
    step = (s_new / (1.0 - rho_1 ** t)) / (1e-12 + np.sqrt(r_new / (1.0 - rho_2 ** t)))
    return x - lr * step, {'s': s_new, 'r': r_new, 't': t + 1}


# Optimiser registry, looked up by name like the reference's gd_func_dict (p2w:362, 527, 705); every value is a
# callable with the reference's convention, `rule(x, grad, lr, **state) -> (x_new, new_state)`.  While an entry IS
# one of the two functions above, the loop runs the same rule on the device (ba_sgd_step / ba_adam_step and the
# fused forms) and never calls it.  Any other callable - a new name, or a wrapper put in place of 'SGD' / 'Adam',
# which may forward to the original - is called once per free group in the order X, C, Theta, phi_x on host copies
# (p2w:704-708); as in the reference (p2w:469-473) its state is {'s', 'r', 't'} under the name 'Adam' and empty
# under any other name.  That is the slow path (parameters and gradients cross PCIe every iteration, nothing is
# queued ahead); with several ranks the rule sees each rank's own block of X.
gd_func_dict = {'SGD': sgd, 'Adam': gd_adam}
_device_rules = {sgd: 1, gd_adam: 2}        # rule numbers of the C ABI

# optimisers beyond the reference's registry, accepted by fit_scene(optimizer=...)
extra_optimizers = {'LM': 'Levenberg-Marquardt with Schur complement (lm.LMFit)'}


class SceneFit:
    """One `fit_scene` run as an object: `step()` executes one pass of the loop body and
    returns False when the loop is over.

    State lives in tensors that are never written in place once they are current: the optimiser
    step returns new ones.  `release` (the snapshot of p2w:397-398) and the stashed Adam moments
    (p2w:411-414) are therefore plain references to earlier tensors."""

    def __init__(self, shard, lr, max_iters, X_mask=True, C_Theta_mask=True, phi_x_mask=True,
                 optimizer='Adam', patience=1000, factor=2.0, stop_value=0.0, stop_diff=1e-12,
                 print_step=1000, ret_all_iters=False, printer=print, speculate=True, fuse=True):
        if optimizer not in gd_func_dict:
            raise ValueError(f'Unkown name of the optmizer, choose one from {[key for key in gd_func_dict]}')
        self.sh = shard
        rule = gd_func_dict[optimizer]
        if not callable(rule):
            raise ValueError(f'gd_func_dict[{optimizer!r}] must be callable as rule(x, grad, lr, **state)')
        self.host_rule = None if rule in _device_rules else rule
        if self.host_rule is not None:
            speculate = fuse = False
        self.lr, self.max_iters = lr, int(max_iters)
        self.masks = (bool(X_mask), bool(C_Theta_mask), bool(phi_x_mask))
        self.optimizer = optimizer
        self.patience, self.factor = patience, factor
        self.stop_value, self.stop_diff, self.print_step = stop_value, stop_diff, print_step
        self.ret_all_iters = ret_all_iters
        self.print = printer
        self.speculate = bool(speculate)   # queue the update before the energy is known
        # let the observation pass apply the X update (only as a queued step: the update is then
        # part of the pass that produces the energy)
        self.fuse = bool(fuse) and self.speculate and bool(X_mask) and bool(getattr(shard, 'can_fuse', False))
        if getattr(shard, 'distributed', False):
            # every rank must walk the same sequence of collectives: fuse only if all of them can
            self.fuse = bool(shard.all_ranks_agree(self.fuse))
        self._grad_x_valid = False         # gX holds the gradient at the current parameters
        self.dropped = 0                   # queued updates the schedule overruled
        self.timed_every = 0        # record HIP events around the observation-pass kernel of every n-th pass
        self._passes = 0
        self.gauge = self.masks[0] and self.masks[1]      # p2w:533, 710

        K6 = 6 * shard.K
        lo = 0 if self.masks[1] else K6
        hi = (K6 + 1) if self.masks[2] else K6
        self.cam_range = (lo, hi) if hi > lo else None     # free slice of [C | Theta | phi_x]

        if self.gauge:
            shard.gauge_fix()                                            # p2w:533-536

        self.release = shard.snapshot()                                  # p2w:463-464
        self.adam = rule is gd_adam
        self.host_state = self.host_state_tmp = None
        if self.host_rule is not None:                                   # p2w:466-473
            self._host_state_named = optimizer
            self.host_state, self.host_state_tmp = self._host_state_init(optimizer), self._host_state_init(optimizer)
        self.mom_x = self.mom_cam = self.mom_x_tmp = self.mom_cam_tmp = None
        if self.adam:                                                    # p2w:469-473
            self.mom_x, self.mom_cam = self._zero_moments()
            self.mom_x_tmp, self.mom_cam_tmp = self._zero_moments()
            self.t = self.t_tmp = 1
        self.all_iters = self.max_iters * [None] if ret_all_iters else None

        self.E = np.zeros(self.max_iters)                                # p2w:592-599
        self.E_min = self.E_min_temp = np.inf
        self.patience_timer, self.success_timer = 0, patience
        self.began_decreasing = self.increased = False
        self.iters = 0
        self.finished = False
        self._intervened = False    # the schedule changed lr / moments / parameters in this pass
        self._ahead = None          # results of a step whose pass was launched from inside the previous wait
        self.launch_ahead = True    # let the wait for the energy launch the next pass (see _arm_next)
        self.fired = 0              # passes launched that way

    # ---- optimiser state moves (p2w:411-414, 439-446) -----------------------------------------
    def _zero_moments(self):
        sh = self.sh
        zx = (sh.X.new_zeros(sh.X.shape), sh.X.new_zeros(sh.X.shape)) if self.masks[0] else None
        zc = (sh.cam.new_zeros(sh.cam.shape), sh.cam.new_zeros(sh.cam.shape)) if self.cam_range else None
        return zx, zc

    def _adam_reset(self):
        if self.adam:
            self.mom_x, self.mom_cam = self._zero_moments()
            self.t = 1
        if self.host_rule is not None:
            self.host_state = self._host_state_init(self._host_state_named)

    def _adam_stash_and_reset(self):
        if self.adam:
            self.mom_x_tmp, self.mom_cam_tmp, self.t_tmp = self.mom_x, self.mom_cam, self.t
        if self.host_rule is not None:
            self.host_state_tmp = self.host_state
        self._adam_reset()

    def _adam_restore(self):
        if self.adam:
            self.mom_x, self.mom_cam, self.t = self.mom_x_tmp, self.mom_cam_tmp, self.t_tmp
        if self.host_rule is not None:
            self.host_state = {k: {n: v.copy() for n, v in st.items()} for k, st in self.host_state_tmp.items()}

    # ---- user-supplied update rule on host copies (p2w:704-708) --------------------------------
    def _host_group_shapes(self):
        N, K = self.sh.N, self.sh.K
        out = []
        if self.masks[0]:
            out.append(('X', (N, 3)))
        if self.masks[1]:
            out.extend([('C', (K, 3)), ('Theta', (K, 3))])
        if self.masks[2]:
            out.append(('phi_x', (1,)))
        return out

    def _host_state_init(self, optimizer):
        """Per-group state as form_params_dict makes it (p2w:466-473): moments and the step counter under
        the name 'Adam', nothing otherwise."""
        if optimizer != 'Adam':
            return {name: {} for name, _ in self._host_group_shapes()}
        return {name: {'s': np.zeros(shape), 'r': np.zeros(shape), 't': np.array([1])}
                for name, shape in self._host_group_shapes()}

    def _host_fetch(self):
        """Host copies of every group's parameters (views of `cam` where they live there) and gradients."""
        sh = self.sh
        K3, inv = 3 * sh.K, 1.0 / sh.nV
        cam = sh.cam.cpu().numpy().copy()
        red = sh.red.cpu().numpy() * inv
        main = {'C': cam[:K3].reshape(-1, 3), 'Theta': cam[K3:2 * K3].reshape(-1, 3), 'phi_x': cam[2 * K3:2 * K3 + 1]}
        grad = {'C': red[:K3].reshape(-1, 3), 'Theta': red[K3:2 * K3].reshape(-1, 3), 'phi_x': red[2 * K3:2 * K3 + 1]}
        if self.masks[0]:
            main['X'] = sh.points_to_caller(sh.X).cpu().numpy().copy()
            grad['X'] = sh.points_to_caller(sh.gX).cpu().numpy() * inv
        return cam, main, grad

    def _host_store(self, cam, main):
        """The updated host copies become the current device state; then the gauge fix on the device."""
        import torch
        sh = self.sh
        X_new = sh.points_to_engine(main['X']) if self.masks[0] else sh.X
        sh.adopt(X_new, torch.from_numpy(cam).to(sh.cam.device))
        sh.invalidate_table()
        if self.gauge:
            sh.gauge_fix()                                               # p2w:710-712

    def _host_update(self):
        """One step of the registered callable on every free group (p2w:704-708)."""
        cam, main, grad = self._host_fetch()
        for name, shape in self._host_group_shapes():
            new, new_state = self.host_rule(main[name], grad[name], self.lr, **self.host_state[name])
            main[name][:] = np.asarray(new, dtype=np.float64).reshape(shape)
            for key, val in self.host_state[name].items():
                val[:] = new_state[key]
        self._host_store(cam, main)

    def _evaluate(self):
        timed = self.timed_every > 0 and self._passes % self.timed_every == 0
        self._passes += 1
        self.sh.eval(grad_x=self.masks[0], grad_cam=self.masks[1], grad_phi=self.masks[2], timed=timed)
        self._grad_x_valid = True

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
            self.release = self.sh.snapshot()            # release <- main (p2w:397-398), by reference
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
                self._intervened = True
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
                self.sh.restore(self.release)         # main <- release (p2w:430-431)
                self._evaluate()          # gradients at the rolled-back parameters
                if (not self.increased) or (self.E_min_temp > self.E_min):
                    self._adam_reset()
                else:
                    self._adam_restore()
                    self.increased = False
                self._intervened = True
        if self.ret_all_iters:
            self.all_iters[entry] = self._free_tensors()

    def _free_tensors(self):
        """The free parameter groups of the current state (X; free slice of [C | Theta | phi_x])."""
        out = []
        if self.masks[0]:
            out.append(self.sh.X.view(-1))
        if self.cam_range:
            out.append(self.sh.cam[self.cam_range[0]:self.cam_range[1]])
        return out

    # ---- one pass of the loop body (p2w:600-714) ----------------------------------------------
    def step(self):
        if self.finished or not (self.iters < self.max_iters):
            self.finished = True
            return False
        armed = None
        if self.fuse:
            queued, self._ahead = (self._ahead, None) if self._ahead is not None else (self._queue_fused(), None)
            armed = self._arm_next(queued)
        else:
            self._evaluate()
            queued = self._queue_update() if self.speculate else None
        fired = 0
        if armed is not None:
            self.E[self.iters], fired = self.sh.wait_energy_fire(*armed['window'])
        else:
            self.E[self.iters] = self.sh.wait_energy()
        old_release, old_iters = self.release, self.iters
        self._intervened = False
        self._manage()
        if fired and (self._intervened or self.iters != old_iters or (self.release is old_release) != (fired == 2)):
            raise RuntimeError('a pass was launched ahead for an energy the schedule did not accept')

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

        if self.host_rule is not None:
            self._host_update()
            self.iters += 1
            return True
        if queued is None or self._intervened:
            if queued is not None:
                self.dropped += 1
            if not self._grad_x_valid:        # the fused pass kept the point gradient to itself
                self._evaluate()
            queued = self._queue_update()
        X, cam, self.mom_x, self.mom_cam, X_gauge = queued
        self.sh.adopt(X, cam, X_gauge)
        if self.adam:
            self.t += 1
        self.iters += 1
        if fired:
            self._ahead = self._finish_ahead(armed, fired)
        return True

    def _arm_next(self, queued):
        """Prepares the pass of the NEXT iteration so that the wait for this iteration's energy can launch it
        itself (Shard.wait_energy_fire): the host then is not on the path between two passes.  It fires only for an
        energy at which the schedule does not intervene: an improvement that neither raises the learning rate nor
        meets a stopping rule (the schedule moves its snapshot forward), or no improvement while patience is left
        (it counts down).  The buffers for the fired pass's results are the ones that are dead in the respective
        case - the moments this iteration's pass read; the old snapshot's points after an improvement, the points
        this iteration's pass read otherwise - i.e. the rotation of buffers the allocator would produce, decided
        before the energy is known.  Returns None when the iteration does not qualify."""
        sh = self.sh
        if not self.launch_ahead or not hasattr(sh, 'arm_eval_update') or self.ret_all_iters or self.iters == 0 \
                or self.iters >= self.max_iters - 1 or self.cam_range is None or not self.gauge:
            return None
        X_q, cam_q, mom_x_q, mom_cam_q, rec_q = queued
        X_cur, X_old = sh._X_buf, self.release[0]
        if X_old is X_cur or X_old is X_q or X_cur is X_q or X_old.shape != X_q.shape:
            return None
        s_out = r_out = None
        if self.adam:
            if self.mom_x is None or self.mom_x_tmp is None or self.mom_x[0] is self.mom_x_tmp[0] \
                    or self.mom_x[0] is mom_x_q[0]:
                return None
            s_out, r_out = self.mom_x
        inf = float('inf')
        prev_min = self.E[:self.iters - 2 * self.patience].min() \
            if self.stop_diff > 0.0 and self.iters > 2 * self.patience else None
        # an improvement: not if it would raise the learning rate, nor into the window of the stopping difference
        hi = self.E_min if self.success_timer > 0 else -inf
        if prev_min is not None:
            hi = min(hi, prev_min - self.stop_diff - 1e-12 * abs(prev_min))
        # no improvement: only while patience is left, and not if the stopping difference is met as things stand
        alt_ok = self.patience_timer > 0 and not self.increased and \
            not (prev_min is not None and 0.0 < prev_min - self.E_min < self.stop_diff)
        if hi == -inf and not alt_ok:
            return None
        bc1 = bc2 = 1.0
        if self.adam:
            bc1, bc2 = 1.0 - RHO_1 ** (self.t + 1), 1.0 - RHO_2 ** (self.t + 1)
        timed = self.timed_every > 0 and self._passes % self.timed_every == 0
        outs = (X_old, s_out, r_out, X_cur if alt_ok else None)
        sh.arm_eval_update(2 if self.adam else 1, self.lr, bc1, bc2, X_q, rec_q, cam_q, mom_x_q, outs,
                           grad_cam=self.masks[1], grad_phi=True, timed=timed)
        return {'window': (self.stop_value, hi, self.E_min if alt_ok else inf), 'outs': outs}

    def _finish_ahead(self, armed, fired):
        """The camera-side step that follows a pass launched ahead (the state it reads is current by now)."""
        self.fired += 1
        self._passes += 1
        self._grad_x_valid = False
        bc1 = bc2 = 1.0
        if self.adam:
            bc1, bc2 = 1.0 - RHO_1 ** self.t, 1.0 - RHO_2 ** self.t
        X_old, s_out, r_out, X_alt = armed['outs']
        cam, mom_cam, rec = self.sh.camera_update(2 if self.adam else 1, self.lr, bc1, bc2, self.cam_range, self.mom_cam,
                                                  self.gauge, finalize=True)
        return (X_old if fired == 1 else X_alt), cam, ((s_out, r_out) if self.adam else None), mom_cam, rec

    def _queue_update(self):
        """Update rule on every free group and gauge fix (p2w:704-712) from the current state into
        new tensors; returns them without making them current."""
        bc1 = bc2 = 1.0
        if self.adam:
            bc1, bc2 = 1.0 - RHO_1 ** self.t, 1.0 - RHO_2 ** self.t
        return self.sh.step_update(2 if self.adam else 1, self.lr, bc1, bc2, self.masks[0], self.cam_range,
                                   self.mom_x, self.mom_cam, self.gauge) + (None,)

    def _queue_fused(self):
        """The same step with the X update inside the observation pass: pass + camera-side step."""
        bc1 = bc2 = 1.0
        if self.adam:
            bc1, bc2 = 1.0 - RHO_1 ** self.t, 1.0 - RHO_2 ** self.t
        rule = 2 if self.adam else 1
        timed = self.timed_every > 0 and self._passes % self.timed_every == 0
        self._passes += 1
        # gauge fix on: the camera step also finishes the pass (energy, camera-side gradient); with several
        # GPUs the raw sums of the pass are added up over the ranks between the two launches
        merge = self.gauge
        X, mom_x = self.sh.eval_update(rule, self.lr, bc1, bc2, self.mom_x, grad_cam=self.masks[1],
                                       grad_phi=merge or self.masks[2], timed=timed, defer_finalize=merge)
        self._grad_x_valid = False
        cam, mom_cam, rec = self.sh.cam_raw, self.mom_cam, None
        if self.cam_range is not None:
            cam, mom_cam, rec = self.sh.camera_update(rule, self.lr, bc1, bc2, self.cam_range, self.mom_cam, self.gauge,
                                                      finalize=merge)
        return X, cam, mom_x, mom_cam, rec

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
        n_groups = (1 if self.masks[0] else 0) + (1 if self.cam_range else 0)
        return [[None if snap is None else snap[k].cpu().numpy() for snap in self.all_iters]
                for k in range(n_groups)]
