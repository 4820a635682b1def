"""`vall_e` -- MI355X-native D3PM codec-token sampler behind the reference's module names."""
