"""Step assignment + stage executor (mirror of ``/root/reference/src/pipeline/__init__.py:1-11``)."""

from .pipeline import (
    InputSupplier,
    LatentSpec,
    PipelineConfig,
    PipelineStage,
    run_pipeline_latents,
    run_single_latent,
)
from .step_assignment import StepRange, assign_steps, assign_steps_balanced, assign_steps_rotating, stage_sizes

__all__ = [
    "StepRange",
    "assign_steps",
    "assign_steps_balanced",
    "assign_steps_rotating",
    "stage_sizes",
    "LatentSpec",
    "PipelineStage",
    "PipelineConfig",
    "InputSupplier",
    "run_single_latent",
    "run_pipeline_latents",
]
