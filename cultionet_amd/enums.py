"""String enums of the reference's public surface (/root/reference/src/cultionet/enums/__init__.py)."""
import enum


class StrEnum(str, enum.Enum):
    def __str__(self) -> str:
        return self.value


class AttentionTypes(StrEnum):
    NATTEN = "natten"
    SPATIAL_CHANNEL = "spatial_channel"


class InferenceNames(StrEnum):
    CLASSES_L2 = "classes_l2"
    CLASSES_L3 = "classes_l3"
    CROP_TYPE = "crop_type"
    DISTANCE = "distance"
    EDGE = "edge"
    CROP = "crop"
    RECONSTRUCTION = "reconstruction"


class LossTypes(StrEnum):
    BOUNDARY = "BoundaryLoss"
    CENTERLINE_DICE = "CLDiceLoss"
    CLASS_BALANCED_MSE = "ClassBalancedMSELoss"
    LOG_COSH = "LogCoshLoss"
    FOCAL_TVERSKY = "FocalTverskyLoss"
    TANIMOTO_COMPLEMENT = "TanimotoComplementLoss"
    TANIMOTO = "TanimotoDistLoss"
    TANIMOTO_COMBINED = "TanimotoCombined"
    TVERSKY = "TverskyLoss"


class ModelNames(StrEnum):
    CLASS_INFO = "classes.info"
    CKPT_NAME = "last.ckpt"
    CKPT_TRANSFER_NAME = "last_transfer.ckpt"
    NORM = "last.norm"


class ModelTypes(StrEnum):
    TOWERUNET = "TowerUNet"


class ResBlockTypes(StrEnum):
    RES = "res"
    RESA = "resa"


class LearningRateSchedulers(StrEnum):
    COSINE_ANNEALING_LR = "CosineAnnealingLR"
    EXPONENTIAL_LR = "ExponentialLR"
    ONE_CYCLE_LR = "OneCycleLR"
    STEP_LR = "StepLR"


class ValidationNames(StrEnum):
    TRUE_CROP = "true_crop"
    TRUE_EDGE = "true_edge"
    TRUE_CROP_AND_EDGE = "true_crop_and_edge"
    TRUE_CROP_OR_EDGE = "true_crop_or_edge"
    TRUE_CROP_TYPE = "true_crop_type"
    MASK = "mask"
