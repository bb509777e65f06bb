from .inference import inference_stereo, init_model, is_image_file, is_pfm_file, load_disp, prepare_data

__all__ = ["init_model", "inference_stereo", "prepare_data", "load_disp", "is_image_file", "is_pfm_file"]
