# Sky-model term producers on the predict path; same module paths as africanus/model/.
