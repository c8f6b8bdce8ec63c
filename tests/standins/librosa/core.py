from efficientat_amd.audio_io import load_audio


def load(path, sr=22050, mono=True, **kwargs):
    return load_audio(path, sr=sr, mono=mono)
