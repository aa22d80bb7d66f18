# bess() / bess.one(): the R front end of the reference package (R/R/bess.R:324-764, R/R/bess.one.R:205-267) on
# top of libbessx.  Same arguments, defaults, checks (and their messages) and the same fields in the returned
# object of class "bess"; the solver behind bessCpp() is the HIP library (R/src/bess_amd_shim.cpp ->
# include/bessx.h: bessx_bessCpp).  The reference repeats one block of glue for each of its four algorithm types
# (PDAS, GPDAS, L0L2, GL0L2); here the differences between them are a small table (.bess_algorithms) and the glue
# exists once.
#
# Written against the reference by reading it; R is not available in the build image, so this file has not been
# executed there: R boundary unpinned by execution (DESIGN.md).

.bess_algorithms <- list(
  # name  = c(code passed to bessCpp as algorithm_type, uses group.index, uses the lambda arguments)
  PDAS  = list(code = 1L, grouped = FALSE, ridge = FALSE),
  GPDAS = list(code = 2L, grouped = TRUE,  ridge = FALSE),
  GL0L2 = list(code = 3L, grouped = TRUE,  ridge = TRUE),
  L0L2  = list(code = 5L, grouped = FALSE, ridge = TRUE))

.bess_default_size <- function(x) min(ncol(x), round(nrow(x) / log(nrow(x))))

bess <- function(x, y, family = c("gaussian", "binomial", "poisson", "cox"), type = c("bss", "bsrr"),
                 method = c("gsection", "sequential", "pgsection", "psequential"),
                 tune = c("gic", "ebic", "bic", "aic", "cv"),
                 s.list, lambda.list = 0,
                 s.min, s.max,
                 lambda.min = 0.001, lambda.max = 100, nlambda = 100,
                 always.include = NULL,
                 screening.num = NULL,
                 normalize = NULL, weight = NULL,
                 max.iter = 20, warm.start = TRUE,
                 nfolds = 5,
                 group.index = NULL,
                 seed = NULL) {
  set.seed(seed)  # as the reference does (R/R/bess.R:337); the C++ side never sees it (folds: INTEGRATION.md 3)
  on.exit(set.seed(NULL), add = TRUE)
  if (missing(s.list)) s.list <- 1:.bess_default_size(x)
  if (missing(s.min)) s.min <- 1
  if (missing(s.max)) s.max <- .bess_default_size(x)

  tune <- match.arg(tune)
  type <- match.arg(type)
  family <- match.arg(family)
  method <- match.arg(method)
  is_cv <- tune == "cv"
  ic_type <- switch(tune, aic = 1L, bic = 2L, gic = 3L, ebic = 4L, cv = 1L)
  model_type <- switch(family, gaussian = 1L, binomial = 2L, poisson = 3L, cox = 4L)
  # method -> (path_type, line search of the Powell path); R/R/bess.R:372-384
  path_type <- if (method == "sequential") 1L else 2L
  line.search <- if (method == "psequential") 2L else 1L

  top <- if (path_type == 1L) s.list[length(s.list)] else s.max
  if (!is.null(group.index)) {
    if (path_type == 1L && top > length(group.index)) stop("The maximum one s.list should not be larger than the number of groups!")
    if (path_type == 2L && top > length(group.index)) stop("s.max is too large. Should be smaller than the number of groups!")
  } else {
    if (path_type == 1L && top > ncol(x)) stop("The maximum one in s.list is too large!")
    if (path_type == 2L && top > ncol(x)) stop("s.max is too large")
  }

  algorithm <- if (!is.null(group.index)) switch(type, bss = "GPDAS", bsrr = "GL0L2")
               else switch(type, bss = "PDAS", bsrr = "L0L2")
  alg <- .bess_algorithms[[algorithm]]
  g_index <- NULL
  g_df <- NULL
  if (alg$grouped) {
    # first column (0-based) of every group in order of appearance, and the group sizes (R/R/bess.R:395-397)
    g_index <- match(unique(group.index), group.index) - 1
    g_df <- c(diff(g_index), length(group.index) - g_index[length(g_index)])
  }
  if (ncol(x) == 1 | is.vector(x)) stop("x should be two columns at least!")

  if (family == "binomial") {
    if (is.factor(y)) y <- as.character(y)
    lev <- unique(y)
    if (length(lev) != 2) stop("Please input binary variable!")
    if (!setequal(lev, c(0, 1))) {  # first value seen -> 0, the other -> 1 (R/R/bess.R:421-428)
      y <- as.numeric(ifelse(y == lev[1], 0, 1))
    }
  }
  if (family == "cox") {
    if (!is.matrix(y)) y <- as.matrix(y)
    if (ncol(y) != 2) stop("Please input y with two columns!")
  }
  if (is.vector(y)) {
    if (nrow(x) != length(y)) stop("Rows of x must be the same as length of y!")
  } else {
    if (nrow(x) != nrow(y)) stop("Rows of x must be the same as rows of y!")
  }
  # normalize -> (is_normal, data_type): NULL = the family's default; the user codes 1/2/3 mean
  # "centre x", "scale x", "both" and map to Data's 2/3/1 (R/R/bess.R:442-467)
  if (is.null(normalize)) {
    is_normal <- TRUE
    data_type <- switch(family, gaussian = 1L, binomial = 2L, poisson = 2L, cox = 3L)
  } else if (normalize != 0) {
    is_normal <- TRUE
    data_type <- if (normalize == 1) 2L else if (normalize == 2) 3L else 1L
  } else {
    is_normal <- FALSE
    data_type <- 0L
  }
  if (!is.matrix(x)) x <- as.matrix(x)
  vn <- colnames(x)
  if (is.null(vn)) vn <- paste("x", 1:ncol(x), sep = "")
  if (is.null(weight)) weight <- rep(1, nrow(x))

  screening <- !is.null(screening.num)
  if (!screening) {
    screening.num <- ncol(x)
  } else {
    if (screening.num > ncol(x)) stop("The number of screening features must be equal or less than that of the column of x!")
    if (path_type == 1L && screening.num < top) stop("The number of screening features must be equal or greater than the maximum one in s.list!")
    if (path_type == 2L && screening.num < top) stop("The number of screening features must be equal or greater than the s.max!")
  }
  if (is.null(always.include)) {
    always.include <- numeric(0)
  } else {
    if (is.na(sum(as.integer(always.include)))) stop("always.include should be an integer vector")
    if (sum(always.include <= 0)) stop("always.include should be an vector containing variable indexes which is possitive.")
    always.include <- as.integer(always.include) - 1
    if (length(always.include) > screening.num) stop("The number of variables in always.include should not exceed the sc")
    if (path_type == 1L && length(always.include) > top) stop("always.include containing too many variables. The length of it should not exceed the maximum in s.list.")
    if (path_type == 2L && length(always.include) > top) stop("always.include containing too many variables. The length of it should not exceed the s.max.")
  }

  # Cox: rows by ascending time, the response handed down is the status column (R/R/bess.R:527-534)
  xs <- x
  ys <- y
  if (model_type == 4L) {
    ord <- order(y[, 1])
    x <- x[ord, , drop = FALSE]
    y <- y[ord, 2]
  }
  # the ridge types take the lambda arguments, the L0 types ignore them (lambda fixed at 0); the plain L0L2
  # sequential path with lambda.list = 0 means the default grid (R/R/bess.R:715)
  if (alg$ridge) {
    if (algorithm == "L0L2" && path_type == 1L && length(lambda.list) == 1 && lambda.list[1] == 0)
      lambda.list <- exp(seq(log(100), log(0.01), length.out = 100))
    lam <- list(seq = lambda.list, min = lambda.min, max = lambda.max)
  } else {
    lam <- list(seq = 0, min = 0, max = 0)
  }
  res <- bessCpp(x, y, data_type = data_type, weight, is_normal = is_normal, algorithm_type = alg$code,
                 model_type = model_type, max_iter = max.iter, exchange_num = 2, path_type = path_type,
                 is_warm_start = warm.start, ic_type = ic_type, is_cv = is_cv, K = nfolds, state = rep(2, 10),
                 sequence = s.list, lambda_seq = lam$seq, s_min = s.min, s_max = s.max, K_max = 10, epsilon = 10,
                 lambda_min = lam$min, lambda_max = lam$max, nlambda = nlambda, is_screening = screening,
                 screening_size = screening.num,
                 powell_path = if (algorithm == "L0L2") line.search else 1,
                 g_index = if (alg$grouped) g_index else (1:ncol(x) - 1),
                 always_select = always.include, tao = 1.1)

  names(res$beta) <- vn
  rename <- c(train_loss = "loss", train_loss_all = "loss.all", beta_all = "beta.all", coef0_all = "coef0.all",
              lambda_all = "lambda.all", ic = if (is_cv) "cvm" else "ic", ic_all = if (is_cv) "cvm.all" else "ic.all")
  hit <- names(res) %in% names(rename)
  names(res)[hit] <- rename[names(res)[hit]]
  # what the fitted object carries (R/R/bess.R:541-556 and the three sibling blocks)
  plain_cox <- algorithm == "PDAS" && family == "cox"
  res$x <- if (plain_cox) xs else x
  res$y <- if (plain_cox) ys else y
  res$family <- family
  res$s.list <- s.list
  res$nsample <- nrow(x)
  res$algorithm_type <- algorithm
  res$method <- method
  res$type <- type
  res$ic.type <- if (is_cv) "cv" else c("AIC", "BIC", "GIC", "EBIC")[ic_type]
  res$s.max <- s.max
  res$s.min <- s.min
  if (alg$grouped) {
    res$group.index <- group.index
    res$g_index <- g_index
    res$g_df <- g_df
  }
  if (alg$ridge) {
    res$lambda.list <- lambda.list
    res$lambda.max <- lambda.max
    res$lambda.min <- lambda.min
    res$nlambda <- nlambda
    if (algorithm == "L0L2") res$line.search <- if (line.search == 1L) "gsection" else "sequential"
  }
  if (screening) res$screening_A <- res$screening_A + 1
  res$call <- match.call()
  class(res) <- "bess"
  # (the reference scatters beta.all from the screened columns back to all p here, recover(); libbessx returns every
  # coefficient vector in the caller's column numbering already)
  # refit of the selected model with the base-R fitters, for the L0 types only (R/R/bess.R:565-577, :631-643)
  if (!alg$ridge) {
    sel <- which(res$beta != 0)
    res$bestmodel <-
      if (family == "gaussian") lm(y ~ x[, sel], weights = weight)
      else if (family == "cox") coxph(Surv(ys[, 1], ys[, 2]) ~ xs[, sel], iter.max = max.iter, weights = weight)
      else glm(y ~ x[, sel], family = family, weights = weight)
  }
  res
}

# One (s, lambda): bess() on a one-point sequential path, then the path-related fields are dropped
# (R/R/bess.one.R:205-267).
bess.one <- function(x, y, family = c("gaussian", "binomial", "poisson", "cox"), type = c("bss", "bsrr"),
                     s, lambda = 0, always.include = NULL,
                     screening.num = NULL,
                     normalize = NULL, weight = NULL,
                     max.iter = 20,
                     group.index = NULL) {
  if (length(s) > 1) stop("bess.one needs only a single value for s.")
  if (length(lambda) > 1) stop("bess.one needs only a single value for lambda.")
  family <- match.arg(family)
  type <- match.arg(type)
  res <- bess(x, y, family = family, type = type, method = "sequential", tune = "gic",
              s.list = s, lambda.list = lambda, s.min = s, s.max = s,
              lambda.min = lambda, lambda.max = lambda, nlambda = 1,
              always.include = always.include, screening.num = screening.num,
              normalize = normalize, weight = weight, max.iter = max.iter, warm.start = TRUE, nfolds = 5,
              group.index = group.index, seed = NULL)
  res$s <- s
  res$bess.one <- TRUE
  res$call <- match.call()
  drop <- c("beta.all", "coef0.all", "loss.all", "ic.all", "lambda.list", "s.list", "ic.type", "s.max", "s.min")
  if (type == "bsrr") {
    drop <- c(drop, "method", "line.search", "lambda.max", "lambda.min", "lambda.all", "nlambda")
    res$algorithm_type <- "L0L2"
  } else {
    res$algorithm_type <- "PDAS"
    res$type <- type
  }
  res[intersect(drop, names(res))] <- NULL
  res
}
