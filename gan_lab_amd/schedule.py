"""Progressive-growing phase machine as pure host logic (no torch, no GPU).

Restates the control flow of ProGANLearner.train (gan_lab/progan/learner.py:451-459, :560-729,
:848, :951-952): per resolution above ``init_res`` one FADE-IN phase then one
STABILISE phase, each ``nimg_transition`` real images long (rounded UP to a multiple of the current
batch size); the image counter advances by ``batch_size`` per D iteration; after the last
resolution's fade-in a FINAL phase of unbounded length starts.  The learner asks ``begin_iter()``
what to do before each main iteration and reports progress with ``after_d_iter()`` /
``end_iter()``; everything numeric (alpha, delta_alpha, batch size) is returned as plain numbers.

Data parallel (new work): ``nimg_transition`` counts REAL IMAGES, and with ``world_size``
ranks one D iteration consumes ``batch_size * world_size`` of them.  ``batch_size`` stays the per-rank
micro-batch the kernels see; the image counter, the round-up of ``nimg_transition`` and ``delta_alpha`` use the
global batch, so a W-rank run walks the same schedule in images as the reference run with a W-times larger batch
(``world_size=1`` is the reference's arithmetic exactly, pinned by the trace of the real learner).
"""
import math

GROW, STABILISE, FINAL = 'grow', 'stabilise', 'final'


def round_nimg_transition(nimg_transition, batch_size):
    """progan/learner.py:451-455 / :646-649."""
    if nimg_transition % batch_size != 0:
        return batch_size * (int(nimg_transition / batch_size) + 1)
    return nimg_transition


def delta_alpha(batch_size, nimg_transition, num_disc_iters):
    """progan/learner.py:653."""
    return batch_size / ((nimg_transition / num_disc_iters) - batch_size)


def ewma_beta(batch_size, gen_bs_mult=1, half_life=10.):
    """progan/learner.py:1124-1127."""
    return .5 ** ((batch_size * gen_bs_mult) / (half_life * 1000.)) if half_life > 0. else 0.


class PhaseSchedule(object):
    def __init__(self, init_res, final_res, bs_dict, nimg_transition_cfg, num_disc_iters=1, world_size=1):
        self.curr_res, self.final_res = init_res, final_res
        self.bs_dict = dict(bs_dict)
        self.nimg_transition_cfg = nimg_transition_cfg
        self.num_disc_iters = num_disc_iters
        self.world_size = int(world_size)
        self.batch_size = self.bs_dict[init_res]
        self.nimg_transition = round_nimg_transition(nimg_transition_cfg, self.global_batch)
        self.nimg_transition_lst = [self.nimg_transition]
        self.curr_img_num = 0
        self.curr_phase_num = 0
        self.progressively_grow = True
        self.fade_in_phase = False
        self.alpha = 1
        self.delta_alpha = None
        self.alpha_tol = 1.e-8

    @property
    def global_batch(self):
        """Real images consumed per D iteration over all ranks."""
        return self.batch_size * self.world_size

    def begin_iter(self):
        """Returns the list of events to apply before this main iteration: any of GROW, STABILISE,
        FINAL (at most one of the first two, then possibly FINAL never together with them)."""
        events = []
        at_boundary = self.curr_img_num == sum(self.nimg_transition_lst)
        if self.curr_res < self.final_res and at_boundary:
            self.curr_phase_num += 1
            if self.curr_phase_num % 2 == 1:
                self.curr_res *= 2
                self.batch_size = self.bs_dict[self.curr_res]
                self.nimg_transition = round_nimg_transition(self.nimg_transition_cfg, self.global_batch)
                self.delta_alpha = delta_alpha(self.global_batch, self.nimg_transition, self.num_disc_iters)
                self.alpha = 0
                self.fade_in_phase = True
                events.append(GROW)
            else:
                events.append(STABILISE)
            self.nimg_transition_lst.append(self.nimg_transition)
        if self.curr_img_num == sum(self.nimg_transition_lst):
            self.curr_phase_num += 1
            self.nimg_transition_lst.append(math.inf)
            self.progressively_grow = False
            events.append(FINAL)
        return events

    def after_d_iter(self):
        self.curr_img_num += self.global_batch

    def end_iter(self):
        """alpha += delta_alpha with the setter's snap-to-1 (base.py:161-170)."""
        if self.fade_in_phase:
            new_alpha = self.alpha + self.delta_alpha
            if not (0. <= new_alpha < 1. + self.alpha_tol):
                raise ValueError('Input alpha parameter must be in the range [0,1].')
            if 1. - self.alpha_tol < new_alpha < 1. + self.alpha_tol:
                self.fade_in_phase = False
                self.alpha = 1
            else:
                self.alpha = new_alpha

    # -- checkpoint / resume ----------------------------------------------------------------------
    def restore(self, curr_res, curr_img_num, curr_phase_num, nimg_transition_lst, alpha, progressively_grow=True):
        """Put the machine into a saved state (the fields ``save_model`` keeps, progan/learner.py:1261-1297)."""
        self.curr_res = int(curr_res)
        self.batch_size = self.bs_dict[self.curr_res]
        self.nimg_transition = round_nimg_transition(self.nimg_transition_cfg, self.global_batch)
        self.nimg_transition_lst = [math.inf if (x is None or x < 0 or x == math.inf) else x
                                    for x in nimg_transition_lst]
        self.curr_img_num, self.curr_phase_num = int(curr_img_num), int(curr_phase_num)
        self.progressively_grow = bool(progressively_grow)
        self.fade_in_phase = alpha != 1
        self.alpha = alpha if self.fade_in_phase else 1
        self.delta_alpha = delta_alpha(self.global_batch, self.nimg_transition, self.num_disc_iters) \
            if self.fade_in_phase else None
        return self
