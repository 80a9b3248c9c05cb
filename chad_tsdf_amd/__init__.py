"""chad_tsdf_amd -- MI355X-native implementation of chad::TSDFMap::insert()."""
