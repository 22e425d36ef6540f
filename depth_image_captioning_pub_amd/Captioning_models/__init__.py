"""Host-side mirror of the reference's `Captioning_models` package for the depth-soft / depth-hard path:
same module paths, class names, constructor signatures, attribute names and state_dict keys, so
`from depth_image_captioning_pub_amd.Captioning_models.Depth_caption_model.depth_models import ...`
is a drop-in for the reference import.  Every forward/backward runs in libdic_hip.so."""
