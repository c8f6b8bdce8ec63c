from efficientat_amd.utils import cnn_out_size, make_divisible  # noqa: F401
