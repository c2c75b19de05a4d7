from .configs import (Config, ControllerConfig, ActionsConfig, RewardConfig, ObservationConfig,  # noqa: F401
                      MemoryConfig, ModelConfig, TrainingConfig, VisuConfig)
