"""Loss side of the reference's problem handlers (``ctunet/pytorch/ProblemHandler.py``).

Only ``comp_losses_metrics`` -- the part of a handler that sits on the hot path
(Model.forward_pass -> comp_losses_metrics, Model.py:363) -- is provided; dataset binding and NIfTI
prediction writers are out of scope.  Class names match the reference so that the
``s_problem_handler`` strings of the example inis resolve here.
"""
from __future__ import annotations

from .losses import comp_losses_metrics_double, comp_losses_metrics_single


class ProblemHandler:
    """ProblemHandler.py:17-102."""
    train_dataset_class = None
    test_dataset_class = None

    comp_losses_metrics = staticmethod(comp_losses_metrics_single)

    def write_predictions(self, predictions, input_filepaths, output_folder_name, input_imgs):
        raise NotImplementedError("ctunet_amd: NIfTI prediction writing is outside the accelerated path")


class ImageTargetProblem(ProblemHandler):
    """ProblemHandler.py:105-163."""


class FlapRec(ImageTargetProblem):
    """ProblemHandler.py:166-176: single-output flap reconstruction."""


class FlapRecWithShapePrior(ImageTargetProblem):
    """ProblemHandler.py:179-189."""


class DenoisingAE(ImageTargetProblem):
    """Single-output handler with the base loss."""


class FlapRecWithShapePriorDoubleOut(ImageTargetProblem):
    """ProblemHandler.py:192-354: (full skull, flap) double output."""

    def __init__(self, with_sp=True):
        self.with_sp = with_sp

    comp_losses_metrics = staticmethod(comp_losses_metrics_double)


class FlapRecDoubleOut(FlapRecWithShapePriorDoubleOut):
    """ProblemHandler.py:357-371."""

    def __init__(self):
        super().__init__(with_sp=False)
